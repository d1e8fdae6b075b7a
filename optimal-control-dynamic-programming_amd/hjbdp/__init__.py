"""hjbdp - host side of libhjbdp, the MI355X-native Bellman-backup path of
abdolrezat/Optimal-Control-Dynamic-Programming.

The classes keep the reference's names and methods so its scripts read the same:

    from hjbdp import Dynamic_Solver
    objA = Dynamic_Solver(); objA.run(); objA.get_optimal_path()

All sweeps run in the HIP library (hjbdp/libhjbdp.so, C ABI in include/hjbdp.h);
there is no CPU fallback.
"""
from . import _abi
from .core import Backup, DeviceBuffer, HjbError, MultiBackup, RankSlab, device_count, device_mem_info, load_library, policy_lookup, solve_batch, solve_many, suggest_axis_order
from .problem import ProblemSpec, Term, permute_state_axes
from .dynamic_solver import Dynamic_Solver
from .solver_position import Solver_position
from .solver_attitude import Solver_attitude
from .solver_pos_att import Solver_pos_att

__all__ = ["Backup", "DeviceBuffer", "device_mem_info", "MultiBackup", "RankSlab", "HjbError", "ProblemSpec", "Term", "permute_state_axes", "Dynamic_Solver", "Solver_position", "Solver_attitude", "Solver_pos_att", "device_count", "load_library", "policy_lookup", "solve_batch", "solve_many", "suggest_axis_order", "_abi"]

function [obj, U_Opt_stages] = Solver_position_hjbdp_simplified_run(obj, varargin)
%SOLVER_POSITION_HJBDP_SIMPLIFIED_RUN  Drop-in body for Solver_position.simplified_run
%   (position-control/Solver_position.m:94-150):
%       sp = Solver_position;  Solver_position_hjbdp_simplified_run(sp);  get_optimal_path(sp)
%   leaves exactly what the reference method leaves: obj.n_mesh_x / obj.n_mesh_v updated to the sym_linspace lengths
%   (:100,:104) and obj.U1_Opt .. obj.U3_Opt = griddedInterpolant({s_x, s_v}, obj.U_vector(U_idx), 'nearest') (:144-146).
%   The mesh lines are the reference's own (:97-104); the [n_x, n_v, 3] tables J_current_i, x_i_next, v_i_next
%   (:108-128) are handed to the library as the 1-D operands they are ndgrid copies of - x_next = X + dx(V),
%   v_next = V + dv(U), J_current = Qx x^2 + Qv v^2 + R u^2, each evaluated left to right as MATLAB does - and the
%   `for k_s = N_stage-1:-1:1` loop (:132-141) runs in libhjbdp (all double, as in the reference).
%   Optional name/value pairs are passed on to hjbdp_solve ('devices', ...).  'n_stages' overrides N_stage - 1.
%   [obj, U_Opt_stages] = ...(obj, 'keep_policy', true) also returns every stage's policy the way the reference's
%   development scripts keep it (attitude-control/test/test_simplified.m:102-104, U1_Opt(:,:,k_s) = U_vector(U1_idx)):
%   U_Opt_stages{ch} is [n_x, n_v, n_stages], stage k_s in page k_s (hjbdp_solve 'keep_stages').
% NOT executed in the build image (no MATLAB); tested twin: hjbdp/solver_position.py (bit-exact against the oracle on
% the reference's 201 x 201 x 3 grid, tests/test_gpu_solvers.py), and this file's call sequence is replayed through
% ctypes by tests/test_gpu_flat_api.py::test_matlab_shim_sequences_solver_position.
    [n_stages, rest, keep_policy] = take_n_stages(obj.N_stage - 1, varargin);
    U_Opt_stages = cell(1, 3);
    if keep_policy, rest = [rest, {'keep_stages', true}]; end
    %% mesh generation (:97-104)
    s_x = {sym_linspace(obj, obj.x_min, obj.x_max, obj.n_mesh_x), ...
           sym_linspace(obj, obj.x_min, obj.x_max, obj.n_mesh_x), ...
           sym_linspace(obj, obj.x_min, obj.x_max, obj.n_mesh_x)};
    obj.n_mesh_x = length(s_x{1});
    s_v = {sym_linspace(obj, obj.v_min, obj.v_max, obj.n_mesh_v), ...
           sym_linspace(obj, obj.v_min, obj.v_max, obj.n_mesh_v), ...
           sym_linspace(obj, obj.v_min, obj.v_max, obj.n_mesh_v)};
    obj.n_mesh_v = length(s_v{1});
    Qx = [obj.Qx1 obj.Qx2 obj.Qx3];  Qv = [obj.Qv1 obj.Qv2 obj.Qv3];  R = [obj.R1 obj.R2 obj.R3];
    U = obj.U_vector;
    T = @(dims, data) struct('dims', dims, 'data', data);
    for ch = 1:3
        x = s_x{ch};  v = s_v{ch};
        % next_stage_states_simplified (:152-186) on the grid VECTORS: RK4_x(X, V, h) = X + h*(k1+2k2+2k3+k4)/6 with the
        % k's functions of V only; RK4_v(V, U, h) = V + h*(k1+2k2+2k3+k4)/6 with every k = U/Mass
        dx = RK4_x(obj, zeros(size(v)), v, obj.h);                 % the V-only part of x_next
        dv = RK4_v(obj, zeros(size(U)), U, obj.h);                 % the U-only part of v_next
        prob = struct();
        prob.knots = {x, v};
        prob.m = numel(U);
        prob.single = false;
        prob.next_terms = {[T(1, x), T(2, dx)], [T(2, v), T(3, dv)]};
        prob.cost_terms = [T(1, Qx(ch) * x.^2), T(2, Qv(ch) * v.^2), T(3, R(ch) * U.^2)];      % J_current (:113)
        out = hjbdp_solve(prob, n_stages, rest{:});                % for k_s = N_stage-1:-1:1 (:132-141)
        U_idx = out.idx;                                           % [n_x, n_v], 1-based, first minimum
        pol = griddedInterpolant({x, v}, U(U_idx), 'nearest');     % :144-146
        if keep_policy, U_Opt_stages{ch} = reshape(U(out.idx_stages), [numel(x), numel(v), n_stages]); end
        switch ch
            case 1, obj.U1_Opt = pol;
            case 2, obj.U2_Opt = pol;
            case 3, obj.U3_Opt = pol;
        end
    end
    fprintf('stage calculation complete!\n')
end

function [n, rest, keep_policy] = take_n_stages(default, args)
    n = default;  rest = args;  keep_policy = false;
    for i = numel(args) - 1:-2:1
        if strcmp(args{i}, 'n_stages'), n = args{i + 1};  rest([i, i + 1]) = [];  end
        if strcmp(args{i}, 'keep_policy'), keep_policy = args{i + 1};  rest([i, i + 1]) = [];  end
    end
end

function [obj, U_Opt_stages] = Solver_attitude_hjbdp_simplified_run(obj, varargin)
%SOLVER_ATTITUDE_HJBDP_SIMPLIFIED_RUN  Drop-in body for Solver_attitude.simplified_run
%   (attitude-control/Solver_attitude.m:196-259):
%       sa = Solver_attitude;  Solver_attitude_hjbdp_simplified_run(sa);
%   leaves obj.U1_Opt .. obj.U3_Opt = griddedInterpolant({s_w, s_t}, obj.U_vector(U_idx), 'nearest') (:249-251).
%   Mesh lines are the reference's (:199-205); the [n_w, n_t, 3] tables (:209-233) go to the library as their 1-D
%   operands - w_next = W + dw(U) (RK4_w, every k = U/J), t_next = T + dt(W) (RK4_t, k's functions of W only),
%   J_current = Qw w^2 + Qt theta^2 + R u^2 (:220) - and the stage loop (:236-247) runs in libhjbdp, in double.
%   Optional name/value pairs are passed on to hjbdp_solve; 'n_stages' overrides N_stage - 1.
%   [obj, U_Opt_stages] = ...(obj, 'keep_policy', true) also returns the policy of EVERY stage as the reference's
%   development script keeps it (attitude-control/test/test_simplified.m:102-104, U1_Opt(:,:,k_s) = U_vector(U1_idx)):
%   U_Opt_stages{ch} is [n_w, n_t, n_stages] with stage k_s in page k_s (hjbdp_solve 'keep_stages': the stage kernel
%   writes every stage's labels straight into that array on the device).
% NOT executed in the build image (no MATLAB); tested twin: hjbdp/solver_attitude.py::simplified_run (bit-exact
% against the oracle at the reference's 1000 x 300 x 3 size, tests/test_gpu_deep.py); call sequence replayed through
% ctypes by tests/test_gpu_flat_api.py::test_matlab_shim_sequences_attitude_simplified.
    n_stages = obj.N_stage - 1;  rest = varargin;  keep_policy = false;  U_Opt_stages = cell(1, 3);
    for i = numel(varargin) - 1:-2:1
        if strcmp(varargin{i}, 'n_stages'), n_stages = varargin{i + 1};  rest([i, i + 1]) = [];  end
        if strcmp(varargin{i}, 'keep_policy'), keep_policy = varargin{i + 1};  rest([i, i + 1]) = [];  end
    end
    if keep_policy, rest = [rest, {'keep_stages', true}]; end
    %% mesh generation (:199-205)
    s_w = linspace(obj.w_min, obj.w_max, obj.n_mesh_w);
    s_t = {linspace(deg2rad(obj.yaw_min), deg2rad(obj.yaw_max), obj.n_mesh_t), ...
           linspace(deg2rad(obj.pitch_min), deg2rad(obj.pitch_max), obj.n_mesh_t), ...
           linspace(deg2rad(obj.roll_min), deg2rad(obj.roll_max), obj.n_mesh_t)};
    Jc = [obj.J1 obj.J2 obj.J3];
    Qw = [obj.Q1 obj.Q2 obj.Q3];  Qt = [obj.Qt1 obj.Qt2 obj.Qt3];  R = [obj.R1 obj.R2 obj.R3];
    U = obj.U_vector;
    T = @(dims, data) struct('dims', dims, 'data', data);
    for ch = 1:3
        t = s_t{ch};
        dw = RK4_w(obj, zeros(size(U)), U, Jc(ch), obj.h);        % :630-644, the U-only part of w_next
        dt = RK4_t(obj, zeros(size(s_w)), s_w, obj.h);            % :646-660, the W-only part of t_next
        prob = struct();
        prob.knots = {s_w, t};                                    % F = griddedInterpolant({s_w, s_t}, ...) (:221)
        prob.m = numel(U);
        prob.single = false;
        prob.next_terms = {[T(1, s_w), T(3, dw)], [T(2, t), T(1, dt)]};
        prob.cost_terms = [T(1, Qw(ch) * s_w.^2), T(2, Qt(ch) * t.^2), T(3, R(ch) * U.^2)];
        out = hjbdp_solve(prob, n_stages, 'fast_axes', false, rest{:});     % :236-247 (the reference's (w, theta) order, as the Python mirror runs it; 'fast_axes', true in rest overrides)
        pol = griddedInterpolant({s_w, t}, U(out.idx), 'nearest');    % :249-251
        if keep_policy, U_Opt_stages{ch} = reshape(U(out.idx_stages), [numel(s_w), numel(t), n_stages]); end   % test_simplified.m:102-104
        switch ch
            case 1, obj.U1_Opt = pol;
            case 2, obj.U2_Opt = pol;
            case 3, obj.U3_Opt = pol;
        end
    end
    fprintf('stage calculation complete... cleaning up...\n...Done!\n')
end

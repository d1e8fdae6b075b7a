function obj = Dynamic_Solver_hjbdp_run(obj)
%DYNAMIC_SOLVER_HJBDP_RUN  Drop-in body for Dynamic_Solver.run (test/Dynamic_Solver.m:66-105):
%   objA = Dynamic_Solver;  Dynamic_Solver_hjbdp_run(objA);  get_optimal_path(objA)
% The grid set-up lines are the reference's own (:69-80); the a_D_M / g_D tables (:184-200) are
% passed as their 1-D operands and the `for k=1:N-1` loop (:86-102) runs in libhjbdp.
% NOT executed in the build image (no MATLAB); tested twin: hjbdp/dynamic_solver.py.
    obj.s_r = single(linspace(obj.x_min, obj.x_max, obj.dx));                   % :69
    [obj.X1_mesh, obj.X2_mesh] = ndgrid(obj.s_r, obj.s_r);                      % :70
    U_mesh = linspace(obj.u_min, obj.u_max, obj.du);                            % :72
    obj.J_star = zeros([size(obj.X1_mesh), obj.N], 'single');                   % :77
    obj.u_star = obj.J_star;                                                    % :80
    T = @(dims, data) struct('dims', dims, 'data', data);
    prob.knots = {double(obj.s_r), double(obj.s_r)};
    prob.m = obj.du;
    prob.single = true;
    prob.next_terms = {[T(1, obj.A(1) * obj.s_r), T(2, obj.A(3) * obj.s_r), T(3, obj.B(1) * U_mesh)], ...   % :186
                       [T(1, obj.A(2) * obj.s_r), T(2, obj.A(4) * obj.s_r), T(3, obj.B(2) * U_mesh)]};      % :187
    prob.cost_terms = [T(1, obj.Q(1) * obj.s_r.^2), T(2, obj.Q(4) * obj.s_r.^2), T(3, obj.R * U_mesh.^2)];  % :198-199
    out = hjbdp_solve(prob, obj.N - 1, 'keep_stages', true);                    % for k = 1:N-1, :86
    idx = reshape(out.idx_stages, obj.dx, obj.dx, obj.N - 1);
    obj.u_star(:, :, 1:obj.N-1) = single(U_mesh(idx));                          % u_star(:,:,k_s) = U_mesh(u_star_idx), :100
    obj.J_star(:, :, 1:obj.N-1) = reshape(out.J_stages, obj.dx, obj.dx, obj.N - 1);
    obj.u_star_idx = out.idx;
    obj.F = griddedInterpolant({obj.s_r, obj.s_r}, out.J, 'linear');           % what the loop leaves in obj.F
end

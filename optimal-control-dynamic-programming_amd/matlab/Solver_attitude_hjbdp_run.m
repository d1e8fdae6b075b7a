function obj = Solver_attitude_hjbdp_run(obj, varargin)
%SOLVER_ATTITUDE_HJBDP_RUN  Drop-in body for Solver_attitude.run (attitude-control/Solver_attitude.m:261-300):
%       sa = Solver_attitude;  Solver_attitude_hjbdp_run(sa);
%   6-D state (w1, w2, w3, yaw, pitch, roll) x 3-D torque.  Leaves what the reference method leaves: obj.F (the
%   interpolant holding J at stage 1, single Values) and obj.U1_Opt .. obj.U3_Opt = single(obj.U_vector(i_k)) over
%   the state grid, (i1, i2, i3) the JOINT minimiser with the cascade's tie-break (min over dims 9, 8, 7, :400-409 =
%   smallest i1, then i2, then i3).  The reference's own post-processing (:290-292) indexes the cascade's outputs
%   linearly and so mixes states up (SURVEY 8a a11); J is unaffected, the labels here are the true minimisers.
%
%   What crosses to the library is what reshape_states (:717-742) prepares - the single typed operands X1V..X3V,
%   cang/sang, U1V..U3V - combined exactly as calculate_J_current_state_fix_shaped (:315-321) and
%   spacecraft_dynamics_taylor_estimate (:413-506) combine them, but never expanded: the cost is nine broadcast
%   operands, w_next three [n_w, n_w, n_u] operands, and the next angles (:449-489: Euler step of the quaternion
%   kinematics, renormalise, atan2 / asin back to angles; independent of the torque) are
%     'on_the_fly', true  (default) computed inside the stage kernel per state from the four [n_q, n_q, n_q]
%                         quaternion tables (hjbdp.h HJB_MODEL_QUAT_EULER321): no nS-sized array exists anywhere, which
%                         is what lets 51^6 states run (176 GB on one MI355X); atan2 / asin are the library's fixed
%                         polynomial forms (3 ulp from libm), so J agrees with the tabulated form to rounding;
%     'on_the_fly', false computed here as the reference does (three nS-sized single arrays) and passed as operands.
%   The library runs the state axes as (yaw, pitch, roll, w1, w2, w3) - the angle axes, whose next value does not depend
%   on the torque, first: its packed stage kernel then contracts them once per state - and the results are permuted back
%   to the reference's dim order here.  'n_stages' overrides N_stage - 1; other name/value pairs go to hjbdp_solve.
% NOT executed in the build image (no MATLAB); tested twin: hjbdp/solver_attitude.py::run / build_spec_model (bit-exact
% against the oracle incl. 24^6 and 51^6 sampled states); call sequence replayed through ctypes by
% tests/test_gpu_flat_api.py::test_matlab_shim_sequences_attitude_run.
    n_stages = obj.N_stage - 1;  on_the_fly = true;  rest = {};
    for i = 1:2:numel(varargin)
        switch varargin{i}
            case 'n_stages',   n_stages = varargin{i + 1};
            case 'on_the_fly', on_the_fly = varargin{i + 1};
            otherwise,         rest = [rest, varargin(i:i + 1)]; %#ok<AGROW>
        end
    end
    obj.reshape_states();                                        % :262, :717-742
    nw = numel(obj.sr_1);  nq = numel(obj.s_yaw);  nu = numel(obj.U_vector);
    X1 = obj.X1V(:);  X2 = obj.X2V(:);  X3 = obj.X3V(:);  UV = obj.U1V(:);          % single
    % quaternion of the grid angles over (yaw, pitch, roll) (:318-320, :419-421)
    C4 = reshape(obj.cang_x4, [], 1, 1);  S4 = reshape(obj.sang_x4, [], 1, 1);
    C5 = reshape(obj.cang_x5, 1, [], 1);  S5 = reshape(obj.sang_x5, 1, [], 1);
    C6 = reshape(obj.cang_x6, 1, 1, []);  S6 = reshape(obj.sang_x6, 1, 1, []);
    x4 = S4 .* C5 .* C6 - C4 .* S5 .* S6;
    x5 = C4 .* S5 .* C6 + S4 .* C5 .* S6;
    x6 = C4 .* C5 .* S6 - S4 .* S5 .* C6;
    x7 = (1 - (x4.^2 + x5.^2 + x6.^2)).^0.5;
    % w_next (:423-425): X1V + h*((J2-J3)/J1*X2V.*X3V + U1V/J1) and its two cyclic partners, each over
    % (two rate axes, one torque dim) in ascending grid-dim order
    h = obj.h;  J1 = obj.J1;  J2 = obj.J2;  J3 = obj.J3;
    A = @(v) reshape(v, [], 1, 1);  Bv = @(v) reshape(v, 1, [], 1);  Cv = @(v) reshape(v, 1, 1, []);
    t1 = h * ((J2 - J3) / J1 * A(X2) .* Bv(X3) + Cv(UV) / J1);   % over (w2, w3, U1)
    t2 = h * ((J3 - J1) / J2 * Bv(X3) .* A(X1) + Cv(UV) / J2);   % over (w1, w3, U2)
    t3 = h * ((J1 - J2) / J3 * A(X1) .* Bv(X2) + Cv(UV) / J3);   % over (w1, w2, U3)
    T = @(dims, data) struct('dims', dims, 'data', data);
    % grid dims as the library runs them: 1 yaw, 2 pitch, 3 roll, 4 w1, 5 w2, 6 w3, 7 U1, 8 U2, 9 U3
    prob = struct();
    prob.knots = cellfun(@(k) double(single(k)), {obj.s_yaw, obj.s_pitch, obj.s_roll, obj.sr_1, obj.sr_2, obj.sr_3}, ...
                         'UniformOutput', false);                % F's grid vectors with single Values: single arithmetic
    prob.m = [nu nu nu];
    prob.single = true;
    wterms = {[T(4, X1), T([5 6 7], t1)], [T(5, X2), T([4 6 8], t2)], [T(6, X3), T([4 5 9], t3)]};
    if on_the_fly
        prob.next_terms = [{[], [], []}, wterms];
        prob.model = struct('h', double(single(h)), 'tables', {{x4, x5, x6, x7}});
    else
        % :449-489 in single over the whole grid (axes yaw, pitch, roll, w1, w2, w3)
        W1 = reshape(X1, 1, 1, 1, [], 1, 1);  W2 = reshape(X2, 1, 1, 1, 1, [], 1);  W3 = reshape(X3, 1, 1, 1, 1, 1, []);
        X4n = x4 + h * (0.5 * (W3 .* x5 - W2 .* x6 + W1 .* x7));
        X5n = x5 + h * (0.5 * (-W3 .* x4 + W1 .* x6 + W2 .* x7));
        X6n = x6 + h * (0.5 * (W2 .* x4 - W1 .* x5 + W3 .* x7));
        X7n = x7 + h * (0.5 * (-W1 .* x4 - W2 .* x5 - W3 .* x6));
        nrm = sqrt(X4n.^2 + X5n.^2 + X6n.^2 + X7n.^2);           % :477
        X4n = X4n ./ nrm;  X5n = X5n ./ nrm;  X6n = X6n ./ nrm;  X7n = X7n ./ nrm;
        yaw_n   = atan2(2 .* (X6n .* X5n + X7n .* X4n), X7n.^2 + X6n.^2 - X5n.^2 - X4n.^2);       % :485-486
        pitch_n = asin(-2 .* (X6n .* X4n - X7n .* X5n));                                         % :487
        roll_n  = atan2(2 .* (X5n .* X4n + X7n .* X6n), X7n.^2 - X6n.^2 - X5n.^2 + X4n.^2);       % :488-489
        prob.next_terms = [{T(1:6, yaw_n), T(1:6, pitch_n), T(1:6, roll_n)}, wterms];
    end
    % J_current_state_fix (:316-321): Q1 X1V^2 + Q2 X2V^2 + Q3 X3V^2 + Q4 x4^2 + Q5 x5^2 + Q6 x6^2 + R1 U1V^2 + R2 U2V^2 + R3 U3V^2
    prob.cost_terms = [T(4, obj.Q1 * X1.^2), T(5, obj.Q2 * X2.^2), T(6, obj.Q3 * X3.^2), ...
                       T(1:3, obj.Q4 * x4.^2), T(1:3, obj.Q5 * x5.^2), T(1:3, obj.Q6 * x6.^2), ...
                       T(7, obj.R1 * UV.^2), T(8, obj.R2 * UV.^2), T(9, obj.R3 * UV.^2)];
    out = hjbdp_solve(prob, n_stages, 'labels', 'auto', rest{:});              % for k_s = N_stage-1:-1:1 (:280-287)
    toref = @(v) permute(reshape(v, [nq nq nq nw nw nw]), [4 5 6 1 2 3]);      % back to (w1, w2, w3, yaw, pitch, roll)
    obj.F = griddedInterpolant({obj.sr_1, obj.sr_2, obj.sr_3, obj.s_yaw, obj.s_pitch, obj.s_roll}, ...
                               single(toref(out.J)), 'linear');                % :267-269 with the final Values
    lab = toref(out.idx) - 1;                                    % control index, U1 fastest (ndgrid order of dims 7, 8, 9)
    i1 = mod(lab, nu);  i2 = mod(floor(lab / nu), nu);  i3 = floor(lab / (nu * nu));
    obj.U1_Opt = single(obj.U_vector(i1 + 1));                   % :296-298
    obj.U2_Opt = single(obj.U_vector(i2 + 1));
    obj.U3_Opt = single(obj.U_vector(i3 + 1));
    fprintf('...Done!\n')
end

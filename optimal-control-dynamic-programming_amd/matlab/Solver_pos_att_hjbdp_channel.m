function out = Solver_pos_att_hjbdp_channel(obj, s_x, s_v, s_t, s_w, f0, f1, f6, f7, Qx, Qv, Qt, Qw, R, J, file_name, varargin)
%SOLVER_POS_ATT_HJBDP_CHANNEL  Drop-in body for Solver_pos_att.calculate_one_channel_U_Opt
%   (pos-att/Solver_pos_att.m:244-297), same argument list:
%       Solver_pos_att_hjbdp_channel(obj, s_x,s_v,s_t,s_w, f0,f1,f6,f7, Qx,Qv,Qt,Qw,R, J, 'channel_x_controller_1')
%   saves exactly what the reference saves (:289): F_gI (griddedInterpolant over {s_x,s_v,s_t,s_w}, single Values = J
%   at the stage the sweep ended on), U_Optimal_id ([n_x,n_v,n_t,n_w] 1-based labels into the *_allcomb vectors) and
%   f0_allcomb .. f7_allcomb (vectors_allcomb, :886-904).
%   Typing is the reference's: the query tables x_next .. w_next are DOUBLE (:299-327) while F_gI.Values and the
%   stage cost are single (:264-265, :800) -> 'double_tables'; the early-stop monitor (:268-285: every 50 stages,
%   tol 1e-2) sums the single array in single -> 'monitor_single'.  The four [n_x,n_v,n_t,n_w,nU] query tables are never
%   formed: x_next = X + h V, v_next = V + h (f1+f2+f6+f7)/Mass, t_next = T + h W, w_next = W + h (moment)/J
%   (:330-402) go to the library as their 1-D operands.
%   'cost_mode' 'exact': J_current_M = single(double sum) exactly as J_current_reshaped (:784-802) forms
%   it, passed as one [n_x,n_v,n_t,n_w,nU] operand (fine up to ~1e8 entries); 'terms': its five separable operands,
%   summed in single inside the library in the reference's order (<= 2 ulp from the double sum) - for grids like 120^4;
%   'f64' (DEFAULT): the separable operands in DOUBLE, summed in double per (state, control) and rounded to single ONCE -
%   bit-identical to 'exact' at any grid size (hjbdp.h HJB_COST_F64; ~10 % slower than 'terms' on 120^4).
%   'fast_axes' (hjbdp_solve's default: true) lets the library run (x, theta, w, v) - its column-sweep kernel, 2 - 4x faster -
%   with results permuted back (J equal to rounding, see hjbdp_solve); 'fast_axes', false runs the reference's own order.
%   Other name/value pairs go to hjbdp_solve.
% NOT executed in the build image (no MATLAB); tested twin: hjbdp/solver_pos_att.py::calculate_one_channel_U_Opt
% (bit-exact against the oracle on the reference's grid incl. monitor and failure mode); call sequence replayed
% through ctypes by tests/test_gpu_flat_api.py::test_matlab_shim_sequences_pos_att_channel.
    cost_mode = 'f64';  n_stages = obj.N_stage - 1;  rest = {};
    for i = 1:2:numel(varargin)
        switch varargin{i}
            case 'cost_mode', cost_mode = varargin{i + 1};
            case 'n_stages',  n_stages = varargin{i + 1};
            otherwise,        rest = [rest, varargin(i:i + 1)]; %#ok<AGROW>
        end
    end
    %% initialization (:253)
    [f0_allcomb, f1_allcomb, f6_allcomb, f7_allcomb] = vectors_allcomb(obj, f0, f1, f6, f7);
    fa = f0_allcomb(:);  fb = f1_allcomb(:);  fc = f6_allcomb(:);  fd = f7_allcomb(:);
    h = obj.h;  d = obj.T_dist;
    % next_stage_states_simplified (:299-328) with the Euler steps (:330-402), double
    dv = h * ((fa + fb + fc + fd) / obj.Mass);
    dw = h * ((fa * d + fb * (-d) + fc * d + fd * (-d)) / J);
    T = @(dims, data) struct('dims', dims, 'data', data);
    prob = struct();
    prob.knots = {s_x, s_v, s_t, s_w};                           % F_gI's grid vectors (:264)
    prob.m = numel(fa);
    prob.single = true;
    prob.next_terms = {[T(1, s_x), T(2, h * s_v)], [T(2, s_v), T(5, dv)], [T(3, s_t), T(4, h * s_w)], [T(4, s_w), T(5, dw)]};
    cu = R * fa.^2 + R * fb.^2 + R * fc.^2 + R * fd.^2;          % :801
    extra = {};
    switch cost_mode
        case 'exact'    % :261-263, :800-801 as is
            prob.cost_terms = T(1:5, J_current_reshaped(obj, s_x, s_v, s_t, s_w, f0_allcomb, f1_allcomb, f6_allcomb, f7_allcomb, ...
                                                        Qx, Qv, Qt, Qw, R));
        case 'terms'    % sum order of :800: Qx x^2 + Qv v^2 + Qw w^2 + Qt t^2 + (R f.^2 ...)
            prob.cost_terms = [T(1, Qx * s_x.^2), T(2, Qv * s_v.^2), T(4, Qw * s_w.^2), T(3, Qt * s_t.^2), T(5, cu)];
        case 'f64'
            prob.cost_terms = [T(1, Qx * s_x.^2), T(2, Qv * s_v.^2), T(4, Qw * s_w.^2), T(3, Qt * s_t.^2), T(5, cu)];
            extra = {'double_cost', true};
        otherwise
            error('hjbdp:arg', 'cost_mode must be exact, terms or f64');
    end
    out = hjbdp_solve(prob, n_stages, 'monitor_period', 50, 'monitor_tol', 1e-2, 'monitor_single', true, ...
                      'double_tables', true, 'labels', 'auto', extra{:}, rest{:});            % :266-286
    if out.stopped_early
        fprintf('sum of errors in the last 50 stages is under tolerance, breaking loop...\n')
    end
    F_gI = griddedInterpolant({s_x, s_v, s_t, s_w}, single(out.J), 'linear');                 %#ok<NASGU>
    U_Optimal_id = out.idx;                                                                   %#ok<NASGU>
    save(file_name, 'F_gI', 'U_Optimal_id', 'f0_allcomb', 'f1_allcomb', 'f6_allcomb', 'f7_allcomb')   % :289
    fprintf('\nstage calculations complete.\n')
end

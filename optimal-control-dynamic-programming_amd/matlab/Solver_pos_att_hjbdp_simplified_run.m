function Solver_pos_att_hjbdp_simplified_run(obj, varargin)
%SOLVER_POS_ATT_HJBDP_SIMPLIFIED_RUN  Drop-in body for Solver_pos_att.simplified_run (pos-att/Solver_pos_att.m:197-242):
%   the reference's own mesh lines and its four calculate_one_channel_U_Opt calls (x, y, z channels + the x channel with
%   thruster 0 failed), each served by Solver_pos_att_hjbdp_channel; writes the same four .mat controller files.
%   Name/value pairs are passed on ('cost_mode', 'fast_axes', 'devices', 'n_stages', ...).
% NOT executed in the build image (no MATLAB); tested twin: hjbdp/solver_pos_att.py::simplified_run.
    s_x1 = sym_linspace(obj, obj.x_min, obj.x_max, obj.n_mesh_x);  s_x2 = s_x1;  s_x3 = s_x1;       % :200-202
    s_v1 = sym_linspace(obj, obj.v_min, obj.v_max, obj.n_mesh_v);  s_v2 = s_v1;  s_v3 = s_v1;       % :204-206
    s_t1 = sym_linspace(obj, deg2rad(obj.theta1_min), deg2rad(obj.theta1_max), obj.n_mesh_t);       % :208-210
    s_t2 = sym_linspace(obj, deg2rad(obj.theta2_min), deg2rad(obj.theta2_max), obj.n_mesh_t);
    s_t3 = sym_linspace(obj, deg2rad(obj.theta3_min), deg2rad(obj.theta3_max), obj.n_mesh_t);
    s_w1 = sym_linspace(obj, obj.w_min, obj.w_max, obj.n_mesh_w);  s_w2 = s_w1;  s_w3 = s_w1;       % :212-214
    Solver_pos_att_hjbdp_channel(obj, s_x1, s_v1, s_t1, s_w1, obj.F_Thr0, obj.F_Thr1, obj.F_Thr6, obj.F_Thr7, ...
        obj.Qx1, obj.Qv1, obj.Qt1, obj.Qw1, obj.R1, obj.J2, 'channel_x_controller_1', varargin{:});              % :217-221
    Solver_pos_att_hjbdp_channel(obj, s_x2, s_v2, s_t2, s_w2, obj.F_Thr2, obj.F_Thr3, obj.F_Thr8, obj.F_Thr9, ...
        obj.Qx2, obj.Qv2, obj.Qt2, obj.Qw2, obj.R2, obj.J3, 'channel_y_controller_1', varargin{:});              % :223-227
    Solver_pos_att_hjbdp_channel(obj, s_x3, s_v3, s_t3, s_w3, obj.F_Thr4, obj.F_Thr5, obj.F_Thr10, obj.F_Thr11, ...
        obj.Qx3, obj.Qv3, obj.Qt3, obj.Qw3, obj.R3, obj.J1, 'channel_z_controller_1', varargin{:});              % :229-233
    % failure mode in x direction (:235-240)
    Solver_pos_att_hjbdp_channel(obj, s_x1, s_v1, s_t1, s_w1, [0], obj.F_Thr1, obj.F_Thr6, obj.F_Thr7, ...
        obj.Qx1, obj.Qv1, obj.Qt1, obj.Qw1, obj.R1, obj.J2, 'channel_x_controller_1_failure', varargin{:});
end

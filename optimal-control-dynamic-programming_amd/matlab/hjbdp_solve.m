function out = hjbdp_solve(prob, n_stages, varargin)
%HJBDP_SOLVE  Backward Bellman sweep on an AMD MI355X through libhjbdp (C ABI: include/hjbdp.h).
%
%   out = hjbdp_solve(prob, n_stages, 'keep_stages', true, 'monitor_period', 50, 'monitor_tol', 1e-2)
%
%   Replaces the stage loops of the reference solvers
%     test/Dynamic_Solver.m:86-102, position-control/Solver_position.m:132-141,
%     attitude-control/Solver_attitude.m:236-247 / :280-287, pos-att/Solver_pos_att.m:270-286
%   prob fields (all MATLAB-native, column-major, double unless stated):
%     knots      cell{D} of grid vectors            m     [1xC] control grid sizes
%     next_terms cell{D} of struct arrays (dims, data)    cost_terms struct array (dims, data)
%                dims = 1-based grid dims the operand varies along (states 1..D, controls D+1..D+C),
%                data = the reshaped operand exactly as the reference builds it before implicit
%                expansion (Solver_pos_att.m:307-314), any singleton dims squeezed out
%     single     logical: run in single (Dynamic_Solver.m:69) or double (test_coder.m) precision
%   out: J (final values), idx (1-based argmin labels), and with keep_stages J_stages / idx_stages
%        [nS x n_stages] with stage k_s in column k_s, stages_done, stopped_early, sweep_ms.
%
%   NOT executed in the build image (no MATLAB there); the tested twin of this file is
%   hjbdp/core.py (ctypes).  See INTEGRATION.md.
    p = inputParser;
    addParameter(p, 'keep_stages', false);
    addParameter(p, 'monitor_period', 0);
    addParameter(p, 'monitor_tol', 0);
    addParameter(p, 'device', 0);
    parse(p, varargin{:});
    o = p.Results;
    if ~libisloaded('libhjbdp')
        here = fileparts(mfilename('fullpath'));
        loadlibrary(fullfile(here, '..', 'hjbdp', 'libhjbdp.so'), fullfile(here, '..', '..', 'include', 'hjbdp.h'));
    end
    D = numel(prob.knots);  C = numel(prob.m);
    if prob.single, cls = 'single'; ptr = 'singlePtr'; dt = 0; else, cls = 'double'; ptr = 'doublePtr'; dt = 1; end
    s = libstruct('hjb_problem');
    s.D = D;  s.C = C;  s.dtype = dt;  s.index_base = 1;
    n = zeros(1, 6, 'int32');  m = zeros(1, 3, 'int32');
    keep = {};
    for a = 1:D
        n(a) = numel(prob.knots{a});
        keep{end+1} = libpointer('doublePtr', double(prob.knots{a}(:))); %#ok<AGROW>
        s.knots{a} = keep{end};
    end
    m(1:C) = int32(prob.m);
    s.n = n;  s.m = m;
    nt = zeros(1, 6, 'int32');
    for a = 1:D
        T = prob.next_terms{a};  nt(a) = numel(T);
        for k = 1:numel(T)
            keep{end+1} = libpointer(ptr, cast(T(k).data(:), cls)); %#ok<AGROW>
            s.next_terms(a, k).mask = uint32(sum(bitshift(1, T(k).dims - 1)));
            s.next_terms(a, k).data = keep{end};
        end
    end
    s.n_next_terms = nt;
    s.n_cost_terms = numel(prob.cost_terms);
    for k = 1:numel(prob.cost_terms)
        keep{end+1} = libpointer(ptr, cast(prob.cost_terms(k).data(:), cls)); %#ok<AGROW>
        s.cost_terms(k).mask = uint32(sum(bitshift(1, prob.cost_terms(k).dims - 1)));
        s.cost_terms(k).data = keep{end};
    end
    h = libpointer('voidPtrPtr');
    st = calllib('libhjbdp', 'hjb_create', s, int32(o.device), h);
    if st ~= 0, error('hjbdp:create', '%s', calllib('libhjbdp', 'hjb_last_error', [])); end
    cleanup = onCleanup(@() calllib('libhjbdp', 'hjb_destroy', h.Value));
    nS = prod(double(n(1:D)));
    so = libstruct('hjb_solve_opts');
    so.n_stages = int32(n_stages);
    so.monitor_period = int32(o.monitor_period);  so.monitor_tol = o.monitor_tol;
    Jf = libpointer(ptr, zeros(nS, 1, cls));  If = libpointer('int32Ptr', zeros(nS, 1, 'int32'));
    so.J_final = Jf;  so.idx_final = If;
    if o.keep_stages
        Js = libpointer(ptr, zeros(nS * n_stages, 1, cls));  Is = libpointer('int32Ptr', zeros(nS * n_stages, 1, 'int32'));
        so.J_stages = Js;  so.idx_stages = Is;
    end
    r = libstruct('hjb_result');
    st = calllib('libhjbdp', 'hjb_solve', h.Value, so, r);
    if st ~= 0, error('hjbdp:solve', '%s', calllib('libhjbdp', 'hjb_last_error', h.Value)); end
    shape = double(n(1:D));  if D == 1, shape = [shape 1]; end
    out.J = reshape(Jf.Value, shape);
    out.idx = reshape(double(If.Value), shape);
    if o.keep_stages
        out.J_stages = reshape(Js.Value, [nS, n_stages]);
        out.idx_stages = reshape(double(Is.Value), [nS, n_stages]);
    end
    out.stages_done = double(r.stages_done);  out.stopped_early = logical(r.stopped_early);
    out.sweep_ms = r.sweep_ms;
end

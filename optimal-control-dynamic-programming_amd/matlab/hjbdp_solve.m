function out = hjbdp_solve(prob, n_stages, varargin)
%HJBDP_SOLVE  Backward Bellman sweep on AMD MI355X GPUs through libhjbdp's FLAT C API (include/hjbdp_matlab.h:
%   primitives, plain arrays and opaque handles only - everything calllib can marshal).
%
%   out = hjbdp_solve(prob, n_stages, 'keep_stages', true, 'monitor_period', 50, 'monitor_tol', 1e-2, 'devices', 0, 'fast_axes', true)
%
%   Replaces the stage loops of the reference solvers
%     test/Dynamic_Solver.m:86-102, position-control/Solver_position.m:132-141,
%     attitude-control/Solver_attitude.m:236-247 / :280-287, pos-att/Solver_pos_att.m:270-286
%   prob fields (all MATLAB-native, column-major):
%     knots      cell{D} of grid vectors            m     [1xC] control grid sizes
%     next_terms cell{D} of struct arrays (dims, data)    cost_terms struct array (dims, data)
%                dims = 1-based grid dims the operand varies along (states 1..D, controls D+1..D+C),
%                data = the reshaped operand exactly as the reference builds it before implicit
%                expansion (Solver_pos_att.m:307-314), any singleton dims squeezed out
%     single     logical: run in single (Dynamic_Solver.m:69) or double (test_coder.m) precision
%     terminal   optional [nS] terminal cost (default zeros, Dynamic_Solver.m:83-84)
%     model      optional struct (h, tables = {x4, x5, x6, x7}): HJB_MODEL_QUAT_EULER321 - the next values of state axes
%                1..3 (yaw, pitch, roll) are computed inside the stage kernel from the four quaternion tables over those
%                axes (Solver_attitude.m:449-489) instead of being passed as nS-sized operands; next_terms{1..3} are empty.
%                D = 6, C = 3, single only.  What makes 51^6 states possible (Solver_attitude_hjbdp_run.m).
%   'devices': a scalar runs on that GPU (hjb_create_from / hjb_solve_flat); a vector [0 1 .. 7] partitions the LAST
%   state axis over those GPUs of this process (hjb_create_multi_from / hjb_solve_multi_flat; no per-stage planes).
%   'fast_axes' (default TRUE, as the Python mirrors' axis_order = "auto"; false = the reference's own axis order, both held to
%   the oracle bit for bit): let the library relabel the state axes when another labelling runs a faster stage kernel
%   (hjb_problem_suggest_order / hjb_problem_permute_axes: Solver_pos_att's (x, v, theta, w) becomes (x, theta, w, v), 3.7x
%   faster on 120^4); terminal cost in and every output are permuted here, so the caller sees its own axis order.
%   CAVEAT: relabelling changes the order in which the 1-D lerps are taken, so J agrees with the reference order only to
%   a few ulp per stage and an argmin label can differ where two controls tie to rounding.  out.axis_order reports the
%   labelling that ran (1-based: entry i = the caller's axis the library ran as its axis i).
%   'double_tables' (default false; with prob.single): next_terms data go to the library as double and every query is
%   formed, located and weighted in double, the weight rounded to single once - the typing of Solver_pos_att.m:299-327
%   (double x_next .. w_next, single F_gI.Values); costs nothing per stage (hjbdp.h HJB_TAB_F64).
%   'double_cost' (default false; with prob.single): cost_terms data go to the library as double and the stage cost of a
%   (state, control) is their ordered sum in double rounded to single ONCE - single(Qx*x.^2 + ... ) of Solver_pos_att.m:800-801
%   without the [n_x,n_v,n_t,n_w,nU] array, bit-identical to passing that array as one term (hjbdp.h HJB_COST_F64).
%   'monitor_single' (default false): the monitor's sum(F.Values(:)) as a single-precision sum (Solver_pos_att.m:274).
%   'labels' (default 'int32'; 'uint8' | 'uint16' | 'auto'): storage class of the argmin labels inside the library and on
%   the way back (hjbdp.h HJB_IDX_*; 'auto' = the narrowest that holds prod(m)): Solver_pos_att's U_Optimal_id has 9 values.
%   out.idx is converted to double either way (MATLAB's min returns double indices).
%   'on_stage' (default []; a function handle, single device): the caller KEEPS ITS OWN `for k` LOOP - the stage loop below runs in
%   MATLAB on device-resident buffers (hjb_device_malloc / hjb_backup_stage_device: one asynchronous launch per stage, J never
%   crosses PCIe between stages, unlike hjb_backup_stage on host arrays) and calls stop = on_stage(k_s) after stage k_s is
%   enqueued (k_s = n_stages .. 1 as in Dynamic_Solver.m:86); a true return value ends the sweep.  No monitor, no keep_stages.
%   out: J (final values), idx (1-based argmin labels), and with keep_stages J_stages / idx_stages
%        [nS x n_stages] with stage k_s in column k_s, stages_done, stopped_early, sweep_ms.
%
%   The same call sequence is exercised through ctypes by tests/test_gpu_flat_api.py (the build image has no MATLAB);
%   where loadlibrary is unavailable, mex/hjbdp_mex.c is the same sequence as a MEX gateway.  See INTEGRATION.md.
    p = inputParser;
    addParameter(p, 'keep_stages', false);
    addParameter(p, 'monitor_period', 0);
    addParameter(p, 'monitor_tol', 0);
    addParameter(p, 'devices', 0);
    addParameter(p, 'fast_axes', true);
    addParameter(p, 'double_tables', false);
    addParameter(p, 'monitor_single', false);
    addParameter(p, 'double_cost', false);
    addParameter(p, 'labels', 'int32');
    addParameter(p, 'on_stage', []);
    parse(p, varargin{:});
    o = p.Results;
    L = 'libhjbdp';
    if ~libisloaded(L)
        here = fileparts(mfilename('fullpath'));
        loadlibrary(fullfile(here, '..', 'hjbdp', 'libhjbdp.so'), fullfile(here, '..', '..', 'include', 'hjbdp_matlab.h'), 'alias', L);
    end
    D = numel(prob.knots);  C = numel(prob.m);
    if prob.single, cls = 'single'; ptr = 'singlePtr'; dt = 0; else, cls = 'double'; ptr = 'doublePtr'; dt = 1; end
    n = cellfun(@numel, prob.knots);
    b = libpointer('voidPtrPtr');
    check(calllib(L, 'hjb_problem_new', int32(D), int32(C), int32(n), int32(prob.m), int32(dt), int32(1), b), [], 'builder');
    bv = b.Value;
    freeb = onCleanup(@() calllib(L, 'hjb_problem_free', bv));
    ncls = cls;                                   % class of the next-state operands as handed to the library
    top = prod(double(prob.m));                   % the largest (1-based) label
    switch o.labels
        case 'int32',  idt = 0;
        case 'uint8',  idt = 1;
        case 'uint16', idt = 2;
        case 'auto',   idt = 3;
        otherwise, error('hjbdp:arg', 'labels must be int32, uint8, uint16 or auto');
    end
    icls = 'int32';                               % class of the label arrays that come back
    if idt == 1 || (idt == 3 && top <= 255), icls = 'uint8'; elseif idt == 2 || (idt == 3 && top <= 65535), icls = 'uint16'; end
    if o.double_tables && ~prob.single, error('hjbdp:arg', 'double_tables is for prob.single = true'); end
    if o.double_tables || idt ~= 0
        check(calllib(L, 'hjb_problem_set_types', bv, int32(idt), int32(o.double_tables)), bv, 'builder');   % HJB_IDX_*, HJB_TAB_F64
    end
    if o.double_tables, ncls = 'double'; end
    ccls = cls;                                   % class of the cost operands as handed to the library
    if o.double_cost
        if ~prob.single, error('hjbdp:arg', 'double_cost is for prob.single = true'); end
        check(calllib(L, 'hjb_problem_set_cost_type', bv, int32(1)), bv, 'builder');          % HJB_COST_F64
        ccls = 'double';
    end
    if isfield(prob, 'model') && ~isempty(prob.model)      % HJB_MODEL_QUAT_EULER321 = 1
        tb = cellfun(@(t) single(t(:)), prob.model.tables, 'UniformOutput', false);
        check(calllib(L, 'hjb_problem_set_model', bv, int32(1), double(prob.model.h), tb{1}, tb{2}, tb{3}, tb{4}), bv, 'builder');
    end
    mask = @(dims) uint32(sum(bitshift(1, dims - 1)));
    for a = 1:D
        check(calllib(L, 'hjb_problem_set_knots', bv, int32(a - 1), double(prob.knots{a}(:)), int32(n(a))), bv, 'builder');
        T = prob.next_terms{a};
        for k = 1:numel(T)           % MATLAB's left-to-right order of the sum, e.g. A(1)*X1 + A(3)*X2 + B(1)*U (:186)
            v = cast(T(k).data(:), ncls);
            check(calllib(L, 'hjb_problem_add_next_term', bv, int32(a - 1), mask(T(k).dims), v, int64(numel(v))), bv, 'builder');
        end
    end
    for k = 1:numel(prob.cost_terms)
        v = cast(prob.cost_terms(k).data(:), ccls);
        check(calllib(L, 'hjb_problem_add_cost_term', bv, mask(prob.cost_terms(k).dims), v, int64(numel(v))), bv, 'builder');
    end
    order = 1:D;                                  % order(i) = the caller's axis that the library runs as axis i
    if o.fast_axes && D > 1
        ord0 = libpointer('int32Ptr', zeros(1, D, 'int32'));  found = libpointer('int32Ptr', int32(0));
        check(calllib(L, 'hjb_problem_suggest_order', bv, ord0, found), bv, 'builder');
        if found.Value
            check(calllib(L, 'hjb_problem_permute_axes', bv, ord0.Value), bv, 'builder');
            order = double(ord0.Value) + 1;
        end
    end
    nS = prod(double(n));
    shape0 = double(n);  if D == 1, shape0 = [shape0 1]; end
    term = [];
    if isfield(prob, 'terminal') && ~isempty(prob.terminal)
        term = cast(prob.terminal(:), cls);
        if D > 1, term = reshape(permute(reshape(term, shape0), order), [], 1); end
    end
    iptr = [icls 'Ptr'];
    Jf = libpointer(ptr, zeros(nS, 1, cls));  If = libpointer(iptr, zeros(nS, 1, icls));
    done = libpointer('int32Ptr', int32(0));  early = libpointer('int32Ptr', int32(0));  ms = libpointer('doublePtr', 0);
    h = libpointer('voidPtrPtr');
    if isscalar(o.devices)
        check(calllib(L, 'hjb_create_from', bv, int32(o.devices), h), bv, 'builder');
        hv = h.Value;
        freeh = onCleanup(@() calllib(L, 'hjb_destroy', hv));
        if o.monitor_single, check(calllib(L, 'hjb_set_option', hv, 'monitor_single', int64(1)), hv, 'handle'); end
        Js = [];  Is = [];
        if o.keep_stages
            Js = libpointer(ptr, zeros(nS * n_stages, 1, cls));  Is = libpointer(iptr, zeros(nS * n_stages, 1, icls));
        end
        if isempty(o.on_stage)
            check(calllib(L, 'hjb_solve_flat', hv, int32(n_stages), int32(o.monitor_period), o.monitor_tol, term, Jf, If, Js, Is, ...
                          done, early, ms), hv, 'handle');
        else
            % the caller's own stage loop on device buffers (the `for k_s = n_stages:-1:1` of Dynamic_Solver.m:86-102)
            if o.keep_stages || o.monitor_period > 0, error('hjbdp:on_stage', 'on_stage runs without keep_stages and monitor'); end
            dev = int32(o.devices);
            jbytes = int64(nS * (4 + 4 * ~prob.single));  ibytes = int64(nS * numel(typecast(cast(0, icls), 'uint8')));
            dJ = {libpointer('voidPtrPtr'), libpointer('voidPtrPtr')};  dI = libpointer('voidPtrPtr');
            check(calllib(L, 'hjb_device_malloc', dev, jbytes, dJ{1}), hv, 'handle');
            free1 = onCleanup(@() calllib(L, 'hjb_device_free', dev, dJ{1}.Value));
            check(calllib(L, 'hjb_device_malloc', dev, jbytes, dJ{2}), hv, 'handle');
            free2 = onCleanup(@() calllib(L, 'hjb_device_free', dev, dJ{2}.Value));
            check(calllib(L, 'hjb_device_malloc', dev, ibytes, dI), hv, 'handle');
            free3 = onCleanup(@() calllib(L, 'hjb_device_free', dev, dI.Value));
            if isempty(term), term = zeros(nS, 1, cls); end
            check(calllib(L, 'hjb_device_copy', dev, dJ{1}.Value, term, jbytes, int32(0)), hv, 'handle');     % HJB_COPY_H2D
            cur = 1;  t0 = tic;  nd = 0;
            for k_s = n_stages:-1:1
                check(calllib(L, 'hjb_backup_stage_device', hv, dJ{cur}.Value, dJ{3 - cur}.Value, dI.Value, []), hv, 'handle');
                cur = 3 - cur;  nd = nd + 1;
                if o.on_stage(k_s), break; end
            end
            check(calllib(L, 'hjb_check_device_status', hv, []), hv, 'handle');       % synchronises; a query that left the grid
            ms.Value = 1e3 * toc(t0);  done.Value = int32(nd);  early.Value = int32(nd < n_stages);
            check(calllib(L, 'hjb_device_copy', dev, Jf, dJ{cur}.Value, jbytes, int32(1)), hv, 'handle');      % HJB_COPY_D2H
            check(calllib(L, 'hjb_device_copy', dev, If, dI.Value, ibytes, int32(1)), hv, 'handle');
        end
    else
        if o.keep_stages, error('hjbdp:multi', 'keep_stages needs a single device'); end
        check(calllib(L, 'hjb_create_multi_from', bv, int32(numel(o.devices)), int32(o.devices), h), bv, 'builder');
        hv = h.Value;
        freeh = onCleanup(@() calllib(L, 'hjb_destroy_multi', hv));
        check(calllib(L, 'hjb_solve_multi_flat', hv, int32(n_stages), int32(o.monitor_period), o.monitor_tol, term, Jf, If, ...
                      done, early, ms), hv, 'multi');
    end
    shape = shape0(order);  if D == 1, shape = shape0; end      % the grid as the library ran it
    back = @(v) ipermute(reshape(v, shape), order);              % ... and back to the caller's axis order
    if D == 1, back = @(v) reshape(v, shape0); end
    out.J = back(Jf.Value);
    out.idx = back(double(If.Value));                 % MATLAB's min returns double indices
    if o.keep_stages
        Jst = reshape(Js.Value, [nS, n_stages]);  Ist = reshape(double(Is.Value), [nS, n_stages]);
        for k = 1:n_stages
            Jst(:, k) = reshape(back(Jst(:, k)), [], 1);  Ist(:, k) = reshape(back(Ist(:, k)), [], 1);
        end
        out.J_stages = Jst;  out.idx_stages = Ist;
    end
    out.axis_order = order;
    out.stages_done = double(done.Value);  out.stopped_early = logical(early.Value);  out.sweep_ms = ms.Value;

    function check(st, obj, kind)
        if st == 0, return; end
        switch kind
            case 'builder', msg = calllib(L, 'hjb_problem_last_error', obj);
            case 'multi',   msg = calllib(L, 'hjb_multi_last_error', obj);
            otherwise,      msg = calllib(L, 'hjb_last_error', obj);
        end
        error('hjbdp:status', '%s (%s)', msg, calllib(L, 'hjb_status_string', int32(st)));
    end
end

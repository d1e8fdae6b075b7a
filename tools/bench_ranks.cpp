// bench_ranks.cpp - SURVEY 8b (iii): a C++ driver of libhjbdp's one-process-per-GPU path, no Python, no torch, no MPI.
//
//   tools/bench_ranks --gpus N --steps K --warmup W [--grid-n 120] [--no-overlap]
//
// The parent forks N children BEFORE anything touches a GPU (it never does itself); child r is rank r on device r.  Every
// rank describes the same problem through the flat builder API (include/hjbdp_matlab.h) - C4: one channel of Solver_pos_att
// (pos-att/Solver_pos_att.m:244-297) on an n^4 sym_linspace grid x the 9 thruster combinations, float64-built query tables,
// uint8 labels, axes (x, theta, w, v) = what bench.py times - creates its slab of the last axis (hjb_rank_create_from),
// joins the RCCL communicator whose id rank 0 publishes in shared memory (hjb_rank_comm_unique_id / hjb_rank_comm_init) and
// runs hjb_rank_sweep: per stage a grouped ncclSend / ncclRecv halo exchange on the library's transfer stream beside the
// interior planes' kernel, boundary strips behind it.  W warm-up stages, then K timed stages between process barriers; the
// MAX over ranks is the time.  Rank 0 prints ONE JSON line with bench.py's keys (metric, value, unit, n_gpus, steps, warmup,
// ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config, checksum_sum_J: comparable with bench.py's).
//
// build:  g++ -O2 -std=c++17 tools/bench_ranks.cpp -Iinclude -Loptimal-control-dynamic-programming_amd/hjbdp -lhjbdp \
//             -Wl,-rpath,'$ORIGIN/../optimal-control-dynamic-programming_amd/hjbdp' -o tools/bench_ranks      (tools/build_bench_ranks.sh)
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hjbdp.h"

namespace {

// MATLAB's linspace (a + i (b - a) / (n - 1), end points exact) and pos-att's sym_linspace (Solver_pos_att.m:906-918: exactly
// n points, the negative side one interval longer when n is even)
std::vector<double> linspace(double a, double b, int n) {
    std::vector<double> y((size_t)n);
    for (int i = 0; i < n; ++i) y[(size_t)i] = a + ((double)i * (b - a)) / (double)(n - 1);
    if (n > 0) { y[0] = a; y[(size_t)n - 1] = b; }
    if (n == 1) y[0] = b;
    return y;
}
std::vector<double> sym_linspace(double a, double b, int n) {
    const int c = (n + 1) / 2;
    std::vector<double> v1 = (n % 2 == 0) ? linspace(a, 0.0, c + 1) : linspace(a, 0.0, c);
    const std::vector<double> v2 = linspace(0.0, b, c);
    v1.insert(v1.end(), v2.begin() + 1, v2.end());
    return v1;
}

struct Shared {
    std::atomic<int> id_ready;
    std::atomic<int> arrived[4];          // barrier generations
    std::atomic<int> failed;
    char id[128];
    double seconds[64];
    double checksum;
    int variant, halo_lo, halo_hi, split;
};

void barrier(Shared *sh, int gen, int world) {
    sh->arrived[gen].fetch_add(1);
    while (sh->arrived[gen].load() < world && !sh->failed.load()) usleep(50);
}

#define CHECK(expr, what)                                                                        \
    do {                                                                                         \
        const int st_ = (expr);                                                                  \
        if (st_ != HJB_OK) {                                                                     \
            fprintf(stderr, "rank %d: %s: %s (%s)\n", rank, #expr, what, hjb_status_string(st_)); \
            sh->failed.store(1);                                                                 \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

int run_rank(Shared *sh, int rank, int world, int steps, int warmup, int n, bool overlap) {
    const double PI = 3.14159265358979323846;
    // Solver_pos_att.m:96-195 (channel x: thrusters 0, 1, 6, 7; inertia J2)
    const double x_max = 0.2, v_max = 0.1, w_max = 2.0 * (PI / 180.0), th_max = 5.0 * (PI / 180.0);      // deg2rad(x) = x * (pi / 180)
    const double Mass = 4.16, J2 = 0.026817 + 0.00150, h = 0.005, Tdist = 9.65e-2, Thr = 0.13;
    const double Qx = 6.0, Qv = 6.0, Qt = 0.5, Qw = 0.5, R = 0.1;
    const std::vector<double> s_x = sym_linspace(-x_max, x_max, n), s_v = sym_linspace(-v_max, v_max, n);
    const std::vector<double> s_t = sym_linspace(-th_max, th_max, n), s_w = sym_linspace(-w_max, w_max, n);
    // vectors_allcomb (:886-904): ndgrid of the four two-level thrusters, first fastest, opposing pairs removed
    std::vector<double> fa, fb, fc, fd;
    for (int i4 = 0; i4 < 2; ++i4)
        for (int i3 = 0; i3 < 2; ++i3)
            for (int i2 = 0; i2 < 2; ++i2)
                for (int i1 = 0; i1 < 2; ++i1) {
                    const double f1 = i1 ? Thr : 0.0, f2 = i2 ? Thr : 0.0, f3 = i3 ? -Thr : 0.0, f4 = i4 ? -Thr : 0.0;
                    if ((f1 > 0 && f3 < 0) || (f2 > 0 && f4 < 0)) continue;
                    fa.push_back(f1); fb.push_back(f2); fc.push_back(f3); fd.push_back(f4);
                }
    const int nU = (int)fa.size();
    std::vector<double> dv((size_t)nU), dw((size_t)nU);
    std::vector<float> cu((size_t)nU);
    for (int u = 0; u < nU; ++u) {
        dv[(size_t)u] = h * ((fa[u] + fb[u] + fc[u] + fd[u]) / Mass);                               // :330-360
        dw[(size_t)u] = h * ((fa[u] * Tdist + fb[u] * (-Tdist) + fc[u] * Tdist + fd[u] * (-Tdist)) / J2);   // :380-402
        cu[(size_t)u] = (float)(R * (fa[u] * fa[u]) + R * (fb[u] * fb[u]) + R * (fc[u] * fc[u]) + R * (fd[u] * fd[u]));      // R*f.^2 (:801)
    }
    auto scaled = [](const std::vector<double> &v, double c) { std::vector<double> o(v); for (double &x : o) x *= c; return o; };
    auto sq32 = [](const std::vector<double> &v, double q) { std::vector<float> o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = (float)(q * (v[i] * v[i])); return o; };      // Q * x.^2, rounded to single once
    // axes as the library's column-sweep kernel wants them (hjb_problem_suggest_order's answer): (x, theta, w, v)
    const int32_t nn[4] = {n, n, n, n}, mm[1] = {nU};
    hjb_builder b = nullptr;
    CHECK(hjb_problem_new(4, 1, nn, mm, HJB_F32, 1, &b), "builder");
    CHECK(hjb_problem_set_types(b, HJB_IDX_AUTO, HJB_TAB_F64), hjb_problem_last_error(b));
    const std::vector<double> *knots[4] = {&s_x, &s_t, &s_w, &s_v};
    for (int a = 0; a < 4; ++a) CHECK(hjb_problem_set_knots(b, a, knots[a]->data(), n), hjb_problem_last_error(b));
    const std::vector<double> hv = scaled(s_v, h), hw = scaled(s_w, h);
    // x+ = X + h V (dims 0, 3);  theta+ = T + h W (1, 2);  w+ = W + dw(u) (2, control = dim 4);  v+ = V + dv(u) (3, 4)
    CHECK(hjb_problem_add_next_term(b, 0, 1u << 0, s_x.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 0, 1u << 3, hv.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 1, 1u << 1, s_t.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 1, 1u << 2, hw.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 2, 1u << 2, s_w.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 2, 1u << 4, dw.data(), nU), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 3, 1u << 3, s_v.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_next_term(b, 3, 1u << 4, dv.data(), nU), hjb_problem_last_error(b));
    // J_current_reshaped (:800): Qx x^2 + Qv v^2 + Qw w^2 + Qt t^2 + R sum f^2, the five operands in that order
    const std::vector<float> cx = sq32(s_x, Qx), cv = sq32(s_v, Qv), cw = sq32(s_w, Qw), ct = sq32(s_t, Qt);
    CHECK(hjb_problem_add_cost_term(b, 1u << 0, cx.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_cost_term(b, 1u << 3, cv.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_cost_term(b, 1u << 2, cw.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_cost_term(b, 1u << 1, ct.data(), n), hjb_problem_last_error(b));
    CHECK(hjb_problem_add_cost_term(b, 1u << 4, cu.data(), nU), hjb_problem_last_error(b));

    hjb_rank r = nullptr;
    CHECK(hjb_rank_create_from(b, rank, rank, world, overlap ? 1 : 0, &r), hjb_problem_last_error(b));
    (void)hjb_problem_free(b);
    int32_t info[10];
    CHECK(hjb_rank_info(r, info), hjb_rank_last_error(r));
    const int owned = info[1] - info[0], planes = owned + info[2] + info[3];
    const int64_t inner = (int64_t)n * n * n;
    // the communicator: rank 0 publishes the id
    if (rank == 0) {
        CHECK(hjb_rank_comm_unique_id(sh->id), hjb_rank_last_error(nullptr));
        sh->variant = info[5]; sh->halo_lo = info[2]; sh->halo_hi = info[3]; sh->split = info[4];
        sh->id_ready.store(1);
    } else {
        while (!sh->id_ready.load() && !sh->failed.load()) usleep(100);
        if (rank == 1) { sh->halo_lo = info[2]; sh->halo_hi = info[3]; sh->split = info[4]; }      // a rank with a lower neighbour
    }
    if (sh->failed.load()) return 1;
    CHECK(hjb_rank_comm_init(r, sh->id), hjb_rank_last_error(r));
    void *dJ[2] = {nullptr, nullptr}, *dI = nullptr;
    const int64_t jb = inner * planes * 4;
    for (int i = 0; i < 2; ++i) CHECK(hjb_device_malloc(rank, jb, &dJ[i]), "J buffer");
    CHECK(hjb_device_malloc(rank, inner * owned * info[8], &dI), "label buffer");
    {
        std::vector<float> zeros((size_t)(inner * planes), 0.0f);        // zero terminal cost (Solver_pos_att.m:264-265)
        for (int i = 0; i < 2; ++i) CHECK(hjb_device_copy(rank, dJ[i], zeros.data(), jb, HJB_COPY_H2D), "upload");
    }
    int32_t done = 0, early = 0, in0 = 1;
    double ms = 0;
    CHECK(hjb_rank_sweep(r, warmup, 0, 0.0, dJ[0], dJ[1], dI, nullptr, &done, &early, &in0, &ms), hjb_rank_last_error(r));
    void *a0 = in0 ? dJ[0] : dJ[1], *a1 = in0 ? dJ[1] : dJ[0];
    barrier(sh, 0, world);
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(hjb_rank_sweep(r, steps, 0, 0.0, a0, a1, dI, nullptr, &done, &early, &in0, &ms), hjb_rank_last_error(r));   // returns synchronised
    barrier(sh, 1, world);
    sh->seconds[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double sums[2] = {0, 0};
    CHECK(hjb_rank_monitor_sums(r, in0 ? a0 : a1, dI, nullptr, sums), hjb_rank_last_error(r));      // all-reduced: the same on every rank
    if (rank == 0) sh->checksum = sums[0];
    barrier(sh, 2, world);
    (void)hjb_device_free(rank, dJ[0]); (void)hjb_device_free(rank, dJ[1]); (void)hjb_device_free(rank, dI);
    (void)hjb_rank_destroy(r);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    int gpus = 1, steps = 50, warmup = 5, n = 120;
    bool overlap = true;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() { return i + 1 < argc ? atoi(argv[++i]) : 0; };
        if (a == "--gpus") gpus = val();
        else if (a == "--steps") steps = val();
        else if (a == "--warmup") warmup = val();
        else if (a == "--grid-n") n = val();
        else if (a == "--no-overlap") overlap = false;
        else { fprintf(stderr, "usage: bench_ranks --gpus N --steps K --warmup W [--grid-n 120] [--no-overlap]\n"); return 2; }
    }
    if (gpus < 1 || gpus > 64 || steps < 1 || n < 8) { fprintf(stderr, "bad arguments\n"); return 2; }
    Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { perror("mmap"); return 1; }
    memset((void *)sh, 0, sizeof(Shared));
    std::vector<pid_t> kids;
    for (int rnk = 0; rnk < gpus; ++rnk) {          // fork BEFORE any GPU call: the parent never initialises HIP
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) _exit(run_rank(sh, rnk, gpus, steps, warmup, n, overlap));
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
    }
    if (bad || sh->failed.load()) { fprintf(stderr, "bench_ranks: a rank failed\n"); return 1; }
    double t = 0;
    for (int rnk = 0; rnk < gpus; ++rnk) t = std::max(t, sh->seconds[rnk]);
    const double states = std::pow((double)n, 4), backups = states * 9.0 * steps;
    printf("{\"metric\": \"bellman_backups_per_s\", \"value\": %.6e, \"unit\": \"backups/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
           "\"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"strong\", \"vs_baseline\": null, \"dtype\": \"f32\", "
           "\"data\": \"synthetic\", \"config\": {\"workload\": \"C4 Solver_pos_att channel x: %dx%dx%dx%d states (x,theta,w,v) x 9 thruster "
           "combinations, float32, float64-built query tables, uint8 argmin, 1 stage per step\", \"states\": %.0f, \"states_per_gpu\": %.0f, "
           "\"controls\": 9, \"stages\": %d, \"sharding\": \"%s\", \"kernel_variant\": %d, \"driver\": \"tools/bench_ranks.cpp (one process per GPU, "
           "RCCL inside libhjbdp: hjb_rank_sweep)\"}, \"checksum_sum_J\": %.17g}\n",
           backups / t, gpus, steps, warmup, t * 1e3 / steps, n, n, n, n, states, states / gpus, steps,
           gpus > 1 ? (std::string("last state axis (v): ") + std::to_string(n / gpus) + " of " + std::to_string(n) + " planes per GPU, halo " +
                       std::to_string(sh->halo_lo) + "/" + std::to_string(sh->halo_hi) + " planes exchanged per stage over RCCL" +
                       (sh->split ? ", overlapped with the interior planes" : "")).c_str()
                    : "none",
           sh->variant, sh->checksum);
    return 0;
}

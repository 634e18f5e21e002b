// device_api.hip -- host side of the device-resident C ABI (group B of reboundx_amd.h):
// argument checks, translation of the op list into kernel programs, regime choice, scratch.
#include "internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace rebxamd;

struct rebxamd_ctx {
    int device = 0;
    std::string err;
    HeadConsts* hc = nullptr;           // four tables: one per stage of a Runge-Kutta update
    int hc_stage = 0;                   // which of them the next head / staging pass fills
    unsigned* status = nullptr;         // warning bits raised by kernels
    long long* ll = nullptr;            // 1 x int64 result slot
    double* partials = nullptr;         // back-reaction partial sums
    size_t partials_cap = 0;
    double* gf = nullptr;               // gr_full workspace
    size_t gf_cap = 0;
    double* gf4 = nullptr;              // pair sums of a first sweep that shared its pass with the potential sums, until a_const exists
    size_t gf4_cap = 0;
    long long gf4_rows[2] = {0, 0}, gf4_n = -1;
    unsigned* gf_flag = nullptr;        // gr_full control words: stop, sweeps run, vote of the sweep just run (+ one spare)
    const unsigned* gf_stop = nullptr;  // caller's stop word for rebxamd_grfull_sweep (rebxamd_grfull_stop_word)
    double* chunk_sums = nullptr;       // second-stage scratch of the back-reaction reduction
    void* hpin = nullptr;               // 512 B of pinned host memory: small results come back through it (no staging copy inside the runtime)
    unsigned* hpin_dev = nullptr;       // the same memory as the device addresses it
    double* whead = nullptr;            // Jacobi state of a head of several massive bodies between Wisdom-Holman steps (8 x 64 doubles)
    double* star = nullptr;             // 6 x 16 doubles: records of body 0 around Wisdom-Holman drifts (two alternate between steps, one for the leading
                                        // half drift; an open run: two more for the leading drifts of alternate steps, one for the closing half drift)
    // batches of Wisdom-Holman steps carried by one pass (star + massless bodies): tables and records of body 0, one per step
    HeadConsts* hc_batch = nullptr;
    double* star_batch = nullptr;
    int wh_batch = 32;                  // steps per pass (1: one pass per step)
    // a Wisdom-Holman run left unsynchronised between calls (rebxamd_wh_run_begin .. rebxamd_wh_run_end)
    struct WhRun { bool open = false; void* stream = nullptr; int64_t k = 0; double dt = 0.; int mode = 0; long long K = 1; rebxamd_view view; std::vector<rebxamd_op> ops; } run;
    double* integ = nullptr;            // stage columns of the force integrators
    size_t integ_cap = 0;
    double* shard = nullptr;            // per-range forces and reactions of the sharded Jacobi/barycentric modify_orbits_forces
    size_t shard_cap = 0;
    double* scan = nullptr;             // workspace of the prefix/suffix-sum passes (gr, Jacobi/barycentric mof, any layout)
    size_t scan_cap = 0;
    // optional timing of the dominant kernels with HIP events on the launch stream
    bool timing = false;
    bool carried_steps = true;  // Wisdom-Holman runs: kick and drift ride in the force pass when the layout allows
    bool fixed_stacks = true;   // stacks with a kernel instance of their own use it
    bool head_one_launch = true;        // Euler / RK2 / RK4 updates of at most 64 bodies as one launch (off with rebxamd_carried_steps(ctx, 0))
    bool grfull_one_launch = true;      // gr_full of at most 256 bodies as one launch (rebxamd_grfull_one_launch)
    bool seq_exact = true;      // reference-order arithmetic: prefix-sum forces on at most REBXAMD_SEQ_MAX_BODIES bodies are walked in the reference's order
    int chain_rounds = 12;      // modify_orbits_direct on long chains: rounds of the fixed-point iteration before the sequential walk takes over
    int chain_last_rounds = 0;  // rounds the last such call took, -1 if it fell back to the sequential walk
    bool fixed_scratch = false; // the buffers above are slices of a fixed array: never grown, never freed
    std::vector<hipEvent_t> ev;         // pairs: start, stop
    size_t ev_used = 0;
};

// Brackets one kernel launch with an event pair when timing is on.
struct TimedLaunch {
    rebxamd_ctx* ctx; hipStream_t st; bool on;
    TimedLaunch(rebxamd_ctx* c, hipStream_t s) : ctx(c), st(s), on(c->timing) {
        if (!on) return;
        if (ctx->ev_used + 2 > ctx->ev.size()) {
            if (ctx->ev.size() >= 16384) { on = false; return; }
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            ctx->ev.push_back(a); ctx->ev.push_back(b);
        }
        (void)hipEventRecord(ctx->ev[ctx->ev_used], st);
    }
    ~TimedLaunch() {
        if (!on) return;
        (void)hipEventRecord(ctx->ev[ctx->ev_used + 1], st);
        ctx->ev_used += 2;
    }
};

static int fail(rebxamd_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

#define HIP_TRY(call)                                                        \
    do {                                                                     \
        hipError_t _e = (call);                                              \
        if (_e != hipSuccess) return fail(ctx, REBXAMD_ERR_HIP, #call, _e);  \
    } while (0)

// While a Wisdom-Holman run is open on a context (rebxamd_wh_run_begin .. rebxamd_wh_run_end) its tables, records of body 0 /
// of the head and table index belong to that run: the entry points that would overwrite them refuse.
#define REFUSE_DURING_RUN(name)                                                                                              \
    do {                                                                                                                     \
        if (ctx->run.open) return fail(ctx, REBXAMD_ERR_ARG, name ": a Wisdom-Holman run is open on this context (rebxamd_wh_run_end closes it)"); \
    } while (0)

// A small result (at most 512 bytes) from device memory to `dst`, waiting for the stream: through
// the context's pinned buffer, so that the copy is one DMA and not a staged one inside the runtime.
static int read_back(rebxamd_ctx* ctx, void* dst, const void* src_dev, size_t bytes, hipStream_t st) {
    if (bytes > 512) return fail(ctx, REBXAMD_ERR_ARG, "read_back: result larger than the pinned buffer");
    HIP_TRY(hipMemcpyAsync(ctx->hpin, src_dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(dst, ctx->hpin, bytes);
    return REBXAMD_OK;
}

extern "C" int rebxamd_create(int device, rebxamd_ctx** out) {
    if (!out) return REBXAMD_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return REBXAMD_ERR_HIP;
    rebxamd_ctx* ctx = new rebxamd_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipMalloc((void**)&ctx->hc, 4 * sizeof(HeadConsts)) != hipSuccess ||
        hipMalloc((void**)&ctx->status, sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void**)&ctx->ll, sizeof(long long)) != hipSuccess ||
        hipMalloc((void**)&ctx->gf_flag, 4 * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void**)&ctx->star, 96 * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&ctx->whead, 3 * 8 * REBXAMD_MAX_HEAD * sizeof(double)) != hipSuccess ||     // Jacobi state + two copies of the state between kick and drift
        hipHostMalloc(&ctx->hpin, 512, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&ctx->chunk_sums, sizeof(double) * REBXAMD_MAX_OPS * 64 * 3) != hipSuccess ||
        hipMemset(ctx->hc, 0, 4 * sizeof(HeadConsts)) != hipSuccess ||
        hipMemset(ctx->status, 0, sizeof(unsigned)) != hipSuccess) {
        rebxamd_destroy(ctx);
        return REBXAMD_ERR_HIP;
    }
    // (a runtime that will not map the pinned buffer for the device: the warning word then travels by a copy command)
    if (hipHostGetDevicePointer((void**)&ctx->hpin_dev, ctx->hpin, 0) != hipSuccess) { ctx->hpin_dev = nullptr; (void)hipGetLastError(); }
    *out = ctx;
    return REBXAMD_OK;
}

extern "C" void rebxamd_destroy(rebxamd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->hc) (void)hipFree(ctx->hc);
    if (ctx->status) (void)hipFree(ctx->status);
    if (ctx->ll) (void)hipFree(ctx->ll);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->gf) (void)hipFree(ctx->gf);
    if (ctx->gf_flag) (void)hipFree(ctx->gf_flag);
    if (ctx->star) (void)hipFree(ctx->star);
    if (ctx->whead) (void)hipFree(ctx->whead);
    if (ctx->hc_batch) (void)hipFree(ctx->hc_batch);
    if (ctx->star_batch) (void)hipFree(ctx->star_batch);
    if (ctx->gf4) (void)hipFree(ctx->gf4);
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
    if (ctx->chunk_sums) (void)hipFree(ctx->chunk_sums);
    if (ctx->scan) (void)hipFree(ctx->scan);
    if (ctx->shard) (void)hipFree(ctx->shard);
    if (ctx->integ) (void)hipFree(ctx->integ);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" int rebxamd_timing_enable(rebxamd_ctx* ctx, int on) {
    if (!ctx) return REBXAMD_ERR_ARG;
    ctx->timing = on != 0;
    ctx->ev_used = 0;
    return REBXAMD_OK;
}

extern "C" int rebxamd_fixed_stacks(rebxamd_ctx* ctx, int on) {
    if (!ctx) return REBXAMD_ERR_ARG;
    ctx->fixed_stacks = on != 0;
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_one_launch(rebxamd_ctx* ctx, int on) {
    if (!ctx) return REBXAMD_ERR_ARG;
    ctx->grfull_one_launch = on != 0;
    return REBXAMD_OK;
}

extern "C" int rebxamd_sequential_exact(rebxamd_ctx* ctx, int on) {
    if (!ctx) return REBXAMD_ERR_ARG;
    ctx->seq_exact = on != 0;
    return REBXAMD_OK;
}

extern "C" int rebxamd_carried_steps(rebxamd_ctx* ctx, int on) {
    if (!ctx) return REBXAMD_ERR_ARG;
    ctx->carried_steps = on != 0;
    ctx->head_one_launch = on != 0;
    return REBXAMD_OK;
}

extern "C" int rebxamd_timing_read(rebxamd_ctx* ctx, double* total_ms, int* launches) {
    if (!ctx || !total_ms || !launches) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_timing_read: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double sum = 0.;
    for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
        HIP_TRY(hipEventSynchronize(ctx->ev[k + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[k], ctx->ev[k + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int)(ctx->ev_used / 2);
    ctx->ev_used = 0;
    return REBXAMD_OK;
}

extern "C" const char* rebxamd_last_error(rebxamd_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

static View make_view(const rebxamd_view* u) {
    View v;
    v.n = u->n;
    v.n_head = u->n_head;
    v.x = u->x; v.y = u->y; v.z = u->z; v.vx = u->vx; v.vy = u->vy; v.vz = u->vz; v.m = u->m;
    v.ax = u->ax; v.ay = u->ay; v.az = u->az;
    v.beta = u->beta; v.Rt = u->R_tides; v.k1 = u->k1; v.ta = u->tau_a; v.te = u->tau_e; v.ti = u->tau_inc;
    return v;
}

static bool aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static bool all_aligned(const View& v) {
    return aligned16(v.x) && aligned16(v.y) && aligned16(v.z) && aligned16(v.vx) && aligned16(v.vy) && aligned16(v.vz) &&
           aligned16(v.m) && aligned16(v.ax) && aligned16(v.ay) && aligned16(v.az) && aligned16(v.beta) && aligned16(v.Rt) &&
           aligned16(v.k1) && aligned16(v.ta) && aligned16(v.te) && aligned16(v.ti);
}

static int ensure(rebxamd_ctx* ctx, double** buf, size_t* cap, size_t bytes, hipStream_t st) {
    if (bytes <= *cap) return REBXAMD_OK;
    if (ctx->fixed_scratch) return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "the request needs more workspace than the fixed scratch array of this entry point holds");
    if (*buf) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipFree(*buf));
        *buf = nullptr; *cap = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc((void**)buf, want));
    *cap = want;
    return REBXAMD_OK;
}

extern "C" int rebxamd_classify(rebxamd_ctx* ctx, const double* m, int64_t n, int64_t* n_head, void* stream) {
    if (!ctx || !n_head || n < 0 || (n > 0 && !m)) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_classify: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_last_massive(m, n, ctx->ll, st));
    long long h = 0;
    { const int rb = read_back(ctx, &h, ctx->ll, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
    *n_head = h < 1 ? 1 : h;
    return REBXAMD_OK;
}

extern "C" int rebxamd_reset_accelerations(rebxamd_ctx* ctx, const rebxamd_view* view, void* stream) {
    if (!ctx || !view || view->n < 0 || (view->n > 0 && (!view->ax || !view->ay || !view->az)))
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_reset_accelerations: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_reset(make_view(view), (hipStream_t)stream));
    return REBXAMD_OK;
}

// the warning word handed to the host's pinned memory and cleared, in one launch (a copy and a fill are two commands).
// (The host waits with hipStreamSynchronize.  Watching a word that this kernel writes last arrive in host memory is
// 5 microseconds faster and NOT safe: kernels queued ahead of this one release their stores at device scope only, so their
// results in pinned host memory can land after that word -- seen as a stale acceleration in 1 of ~1000 calls.)
__global__ void hand_over_warnings(unsigned* status, unsigned* host_word) {
    host_word[0] = status[0];
    status[0] = 0u;
}

extern "C" int rebxamd_fetch_warnings(rebxamd_ctx* ctx, uint32_t* bits, void* stream) {
    if (!ctx || !bits) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_fetch_warnings: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->hpin_dev) {
        hipLaunchKernelGGL(hand_over_warnings, dim3(1), dim3(1), 0, st, ctx->status, ctx->hpin_dev);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(ctx->hpin, ctx->status, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemsetAsync(ctx->status, 0, sizeof(unsigned), st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    unsigned h = 0;
    memcpy(&h, ctx->hpin, sizeof(h));
    *bits = h;
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ streaming ops

static bool has_backreaction(const rebxamd_op& o) {
    switch (o.kind) {
    case REBXAMD_OP_J2: case REBXAMD_OP_J4: case REBXAMD_OP_GR_POTENTIAL: case REBXAMD_OP_CENTRAL: case REBXAMD_OP_TIDES:
        return true;
    case REBXAMD_OP_MOF:
        return o.iarg == REBX_COORDINATES_PARTICLE;
    default:
        return false;
    }
}

// Runs ops[lo,hi) (no gr_full inside, at most one gr and one centre-of-mass mof) as one program.
// target_lo < 0: the normal path.  target_lo >= 0 (sharded runs): forces act on particles
// [target_lo, n) only, every layout goes through the reduction path, and the per-op pushes on
// the sources are stored in sums_out (device, 3 doubles per op of the run) instead of applied.
// step != nullptr (Wisdom-Holman runs): when the layout takes the head + massless-tail path, body
// 0 is recorded and drifted after the head pass and the tail pass carries the kick and the drift
// of every other body (no acceleration traffic); any other layout returns STEP_NOT_CARRIED untouched.
static constexpr int STEP_NOT_CARRIED = -1000;
// ops[lo,hi) as the program the kernels walk, checked against the columns of the view.  max_src <- the largest source index.
static int build_program(rebxamd_ctx* ctx, const rebxamd_view* u, const View& v, const rebxamd_op* ops, int lo, int hi, Program& prog, int& max_src_out) {
    memset(&prog, 0, sizeof(prog));
    prog.nops = hi - lo;
    prog.G = u->G;
    prog.gr_op = -1;
    prog.mofcom_op = -1;
    bool need_m = false;
    int max_src = -1;
    for (int k = lo; k < hi; k++) {
        const rebxamd_op& o = ops[k];
        DevOp& d = prog.ops[k - lo];
        d.kind = o.kind; d.src = o.src; d.iarg = o.iarg; d.slot = -1; d.a = o.a; d.b = o.b; d.c = 0.; d.d = 0.;
        switch (o.kind) {
        case REBXAMD_OP_RADIATION:
            d.c = 1. / o.a;
            prog.need_vel = 1; prog.need_beta = 1; need_m = true;
            break;
        case REBXAMD_OP_J2: case REBXAMD_OP_J4: case REBXAMD_OP_CENTRAL:
            need_m = true;
            break;
        case REBXAMD_OP_GR:
            d.src = 0;
            d.c = o.a * o.a; d.d = 1. / (o.a * o.a);
            prog.need_vel = 1; need_m = true;
            prog.gr_op = k - lo;
            break;
        case REBXAMD_OP_GR_POTENTIAL:
            d.src = 0;
            d.c = o.a * o.a; d.d = 1. / (o.a * o.a);
            need_m = true;
            break;
        case REBXAMD_OP_TIDES:
            prog.need_tides = 1; need_m = true;
            break;
        case REBXAMD_OP_MOF:
            prog.need_vel = 1; prog.need_tau = 1; need_m = true;
            if (o.iarg == REBX_COORDINATES_PARTICLE) { /* src = reference particle */ }
            else if (o.iarg == REBX_COORDINATES_JACOBI || o.iarg == REBX_COORDINATES_BARYCENTRIC) { d.src = -1; prog.mofcom_op = k - lo; }
            else return fail(ctx, REBXAMD_ERR_ARG, "modify_orbits_forces: unknown coordinates");
            break;
        default:
            return fail(ctx, REBXAMD_ERR_ARG, "unknown op kind");
        }
        if (d.src >= 0) {
            if (d.src >= v.n) return fail(ctx, REBXAMD_ERR_ARG, "op source index out of range");
            if (d.src > max_src) max_src = d.src;
        } else if (!(o.kind == REBXAMD_OP_MOF && o.iarg != REBX_COORDINATES_PARTICLE)) {
            return fail(ctx, REBXAMD_ERR_ARG, "op needs a source index");
        }
    }
    max_src_out = max_src;
    if (v.n <= 0) return REBXAMD_OK;
    if (!v.x || !v.y || !v.z || !v.ax || !v.ay || !v.az) return fail(ctx, REBXAMD_ERR_ARG, "position/acceleration columns missing");
    if (prog.need_vel && (!v.vx || !v.vy || !v.vz)) return fail(ctx, REBXAMD_ERR_ARG, "velocity columns missing");
    if (need_m && !v.m) return fail(ctx, REBXAMD_ERR_ARG, "mass column missing");
    return REBXAMD_OK;
}

static int run_streaming(rebxamd_ctx* ctx, const rebxamd_view* u, const rebxamd_op* ops, int lo, int hi, int mode, hipStream_t st,
                         long long target_lo = -1, double* sums_out = nullptr, bool pushes_only = false, const StepArgs* step = nullptr) {
    const View v = make_view(u);
    Program prog;
    int max_src = -1;
    { const int rcp = build_program(ctx, u, v, ops, lo, hi, prog, max_src); if (rcp != REBXAMD_OK) return rcp; }
    if (v.n <= 0) return REBXAMD_OK;

    const bool exact = (mode == REBXAMD_MODE_EXACT);
    // (a pass that carries an integrator's round also makes 16-byte accesses to the caller's columns in step->aux)
    const bool vec2 = all_aligned(v) && (!step || (aligned16(step->aux[0]) && aligned16(step->aux[1]) && aligned16(step->aux[2])));
    const long long K = v.n_head < v.n ? v.n_head : v.n;
    const bool sparse = target_lo < 0 && (u->tail_massless != 0 || K == v.n) && K >= 1 && K <= REBXAMD_MAX_HEAD && max_src < K;

    const bool wh = step && (step->epi == EPI_WH || step->epi == EPI_WHB);
    const bool applies = step && step->epi == EPI_APPLY;      // the pass also applies a prefix-sum force (any-mass-layout pass only)
    if (step && !applies && !(sparse && (wh ? (K == 1 || step->wh_head == (int)K) : v.n > K))) return STEP_NOT_CARRIED;
    if (applies && (sparse || prog.gr_op >= 0 || prog.mofcom_op >= 0 || target_lo >= 0 || pushes_only)) return STEP_NOT_CARRIED;
    HeadConsts* const hc = ctx->hc + ctx->hc_stage;
    if (sparse) {
        if (step && !wh) {
            // force-integrator update of the bodies behind the head: the head passes of every stage have
            // run (tables ctx->hc[0..]); this pass evaluates all stages per body
            TimedLaunch tl(ctx, st);
            HIP_TRY(launch_stream(v, prog, K, ctx->hc, nullptr, nullptr, false, exact, vec2, ctx->fixed_stacks, ctx->status, st, step));
            return REBXAMD_OK;
        }
        // in a Wisdom-Holman run the head pass (one body) also records and drifts that body
        // (unless the previous step's pass has already filled this step's tables)
        if (wh && step->wh_head > 1) {
            // star + planets: the head pass leaves the additional accelerations of the head bodies in their columns and the
            // tables of this step; the head's own Wisdom-Holman step (gravity among its bodies, Jacobi kick, drift to the
            // next force evaluation) follows and records the frame of the massless bodies for the pass below -- unless the
            // pass of the step before has already done both in its extra workgroup
            // (lead_in_head: the first step of a run -- the start of the run rides in the same launch)
            double* const lead = step->lead_in_head ? const_cast<double*>(step->star0) : nullptr;
            if (step->nsub > 1) {
                // a batch of steps in one pass: the head passes and head steps of all of them first, in one wavefront (unless the
                // pass of the batch before has made them in its extra workgroup)
                if (!step->head_ready) HIP_TRY(launch_wh_head_batch(v, prog, const_cast<HeadConsts*>(step->tabs), ctx->status, K, step->dtk, ctx->whead, const_cast<double*>(step->star),
                                             step->nsub, step->h_sub, step->h, exact, st, lead, step->h0));
            } else if (!step->head_ready) {
                // head pass and head step of this one step in one launch
                const double hh = step->head_h != 0. ? step->head_h : step->h;       // (an unsynchronised run: the grains wait, the head drifts)
                HIP_TRY(launch_wh_head_batch(v, prog, hc, ctx->status, K, step->dtk, ctx->whead, const_cast<double*>(step->star), 1, hh, hh, exact, st,
                                             lead, step->h0, step->pre));
            }
        } else if (wh && step->nsub > 1) {
            // a batch of steps in one pass: the head passes of all of them first, one after the other in one wavefront (unless
            // the pass of the batch before has made them in its extra workgroup)
            if (!step->head_ready) HIP_TRY(launch_head_batch(v, prog, const_cast<HeadConsts*>(step->tabs), const_cast<double*>(step->star), step->nsub, step->h_sub, step->h,
                                      step->lead_in_head ? const_cast<double*>(step->star0) : nullptr, step->h0, st));
        } else if (!(wh && step->head_ready))
            HIP_TRY(launch_head(v, prog, hc, ctx->status, exact, st, wh ? const_cast<double*>(step->star) : nullptr, wh ? step->h : 0.,
                                (wh && step->lead_in_head) ? const_cast<double*>(step->star0) : nullptr, wh ? step->h0 : 0.));
        if (v.n > K) {
            TimedLaunch tl(ctx, st);
            HIP_TRY(launch_stream(v, prog, K, (wh && step->nsub > 1) ? step->tabs : hc, nullptr, nullptr, false, exact, vec2, ctx->fixed_stacks, ctx->status, st, step));
        }
        return REBXAMD_OK;
    }
    // general regime
    if (prog.gr_op >= 0 || prog.mofcom_op >= 0) {
        // massive bodies anywhere: gr and centre-of-mass mof become prefix/suffix-sum passes of
        // their own; the ops around them keep streaming (no op reads accelerations, so only the
        // order of additions into a_i is affected, and it is kept)
        if (target_lo >= 0)
            return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "gr and Jacobi/barycentric modify_orbits_forces on particle shards that hold massive bodies run through the rebxamd_shard_* entry points");
        const int rcw = ensure(ctx, &ctx->scan, &ctx->scan_cap, scan_workspace_doubles(v.n) * sizeof(double), st);
        if (rcw != REBXAMD_OK) return rcw;
        if (!exact) {
            // default arithmetic: the streaming ops run as ONE pass ahead of the prefix-sum ops (one
            // pass over the particles less when a prefix-sum op sits in the middle of the stack; the
            // order in which the terms reach a_i changes, which the rounding-level bar of this mode covers)
            rebxamd_op rest[REBXAMD_MAX_OPS];
            int nrest = 0, nreact = 0, last_scan = -1;
            for (int k = lo; k < hi; k++) {
                if (k - lo != prog.gr_op && k - lo != prog.mofcom_op) { nreact += has_backreaction(ops[k]) ? 1 : 0; rest[nrest++] = ops[k]; }
                else last_scan = k;
            }
            // When the streaming pass is the any-mass-layout one with aligned columns, it also applies
            // the last prefix-sum force: the accelerations make one round trip instead of two.
            const bool rides = ctx->carried_steps && nrest > 0 && nreact > 0 && vec2 && stream_bodies_per_block() == scan_segment_bodies();
            StepArgs sa = {0., 0., nullptr, EPI_APPLY, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr, nullptr, 0., {}};
            if (nrest > 0 && !rides) {
                const int rc = run_streaming(ctx, u, rest, 0, nrest, mode, st);
                if (rc != REBXAMD_OK) return rc;
            }
            for (int k = lo; k < hi; k++) {
                if (k - lo != prog.gr_op && k - lo != prog.mofcom_op) continue;
                TimedLaunch tl(ctx, st);
                HIP_TRY(launch_scan_op(v, prog.ops[k - lo], prog.G, exact, ctx->scan, ctx->status, st, (rides && k == last_scan) ? &sa.tail : nullptr));
            }
            if (rides) {
                const int rc = run_streaming(ctx, u, rest, 0, nrest, mode, st, -1, nullptr, false, &sa);
                if (rc != REBXAMD_OK) return rc == STEP_NOT_CARRIED ? fail(ctx, REBXAMD_ERR_UNSUPPORTED, "streaming pass refused to apply the prefix-sum force") : rc;
            }
            return REBXAMD_OK;
        }
        int a = lo;
        for (int k = lo; k <= hi; k++) {
            const bool scan = k < hi && (k - lo == prog.gr_op || k - lo == prog.mofcom_op);
            if (k < hi && !scan) continue;
            if (k > a) {
                const int rc = run_streaming(ctx, u, ops, a, k, mode, st);
                if (rc != REBXAMD_OK) return rc;
            }
            if (scan) {
                TimedLaunch tl(ctx, st);
                // reference-order arithmetic: with at most REBXAMD_SEQ_MAX_BODIES bodies up to the last massive one the recurrences
                // are walked in the reference's own order (the checker's bits on every body); beyond that, parallel prefix sums
                // (massless bodies behind the last massive one add zeros to the recurrences: the chains stop there)
                const long long nseq = (u->tail_massless != 0 && K >= 1) ? K : v.n;
                if (nseq <= REBXAMD_SEQ_MAX_BODIES && ctx->seq_exact) HIP_TRY(launch_seq_op(v, prog.ops[k - lo], prog.G, ctx->scan, ctx->status, st, nseq));
                else HIP_TRY(launch_scan_op(v, prog.ops[k - lo], prog.G, exact, ctx->scan, ctx->status, st));
            }
            a = k + 1;
        }
        return REBXAMD_OK;
    }
    int nslots = 0;
    for (int k = lo; k < hi; k++)
        if (has_backreaction(ops[k])) prog.ops[k - lo].slot = nslots++;
    prog.nslots = nslots;
    if (pushes_only) {
        HIP_TRY(launch_add_pushes(v, prog, sums_out, st));
        return REBXAMD_OK;
    }
    const long long first = target_lo < 0 ? 0 : target_lo;
    if (first >= v.n) {
        if (sums_out) HIP_TRY(hipMemsetAsync(sums_out, 0, sizeof(double) * 3 * (size_t)prog.nops, st));
        return REBXAMD_OK;
    }
    const int nb = stream_block_count(first, v.n, vec2);
    if (nslots > 0) {
        const int rc = ensure(ctx, &ctx->partials, &ctx->partials_cap, (size_t)nslots * (size_t)nb * 3 * sizeof(double), st);
        if (rc != REBXAMD_OK) return rc;
    }
    int nblocks = 0;
    HIP_TRY(launch_stage(v, prog, hc, st));
    {
        TimedLaunch tl(ctx, st);
        HIP_TRY(launch_stream(v, prog, first, hc, ctx->partials, &nblocks, nslots > 0, exact, vec2, ctx->fixed_stacks, ctx->status, st, applies ? step : nullptr));
    }
    // reference-order arithmetic with at most REBXAMD_SEQ_MAX_BODIES bodies up to the last massive one: the sources collect the
    // pushes one by one in the order the reference visits the particles, instead of as sums per workgroup
    const long long nchain = (u->tail_massless != 0 && K >= 1) ? K : v.n;
    if (exact && nslots > 0 && target_lo < 0 && !sums_out && !step && ctx->seq_exact && nchain <= REBXAMD_SEQ_MAX_BODIES)
        HIP_TRY(launch_source_replay(v, prog, hc, nchain, st));
    else if (nslots > 0 || sums_out) HIP_TRY(launch_finalize(v, prog, ctx->partials, nblocks, ctx->chunk_sums, sums_out, st));
    return REBXAMD_OK;
}

// rebxamd_apply without the argument checks: ops[0, nops) split into streaming runs and gr_full evaluations
static int apply_ops(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode, hipStream_t st) {
    int lo = 0;
    while (lo < nops) {
        if (ops[lo].kind == REBXAMD_OP_GR_FULL) {
            const int rc = rebxamd_grfull(ctx, view, ops[lo].a, ops[lo].iarg, mode, nullptr, (void*)st);
            if (rc != REBXAMD_OK) return rc;
            lo++;
            continue;
        }
        int hi = lo, ngr = 0, ncom = 0;
        while (hi < nops && hi - lo < REBXAMD_MAX_OPS && ops[hi].kind != REBXAMD_OP_GR_FULL) {
            if (ops[hi].kind == REBXAMD_OP_GR && ++ngr > 1) break;
            if (ops[hi].kind == REBXAMD_OP_MOF && ops[hi].iarg != REBX_COORDINATES_PARTICLE && ++ncom > 1) break;
            hi++;
        }
        const int rc = run_streaming(ctx, view, ops, lo, hi, mode, st);
        if (rc != REBXAMD_OK) return rc;
        lo = hi;
    }
    return REBXAMD_OK;
}

extern "C" int rebxamd_apply(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    REFUSE_DURING_RUN("rebxamd_apply");
    if (!view || nops < 0 || (nops > 0 && !ops) || view->n < 0 || view->n_head < 0) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_apply: bad argument");
    if (mode != REBXAMD_MODE_FAST && mode != REBXAMD_MODE_EXACT) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_apply: bad mode");
    HIP_TRY(hipSetDevice(ctx->device));
    return apply_ops(ctx, view, ops, nops, mode, (hipStream_t)stream);
}

// ------------------------------------------------------------------ rebxamd_apply for callers that cannot allocate
//
// Same kernels as rebxamd_apply; the small tables, the warning word and the partial sums of the
// pushes on source particles live in `scratch`, a fixed array of this library's code object, instead
// of buffers of a context.  Each GPU holds its own copy of the array at its own address: the address
// is looked up once per device (hipGetSymbolAddress, with that device current) and only that pointer
// is handed to kernels and memsets -- the array's name is a host-side stand-in, not a device address.
static constexpr size_t STATIC_SCRATCH_BYTES = 4u << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

static constexpr size_t round256(size_t b) { return (b + 255) & ~(size_t)255; }
static constexpr size_t SS_TABLES = 0;                                                   // 4 x HeadConsts
static constexpr size_t SS_STATUS = SS_TABLES + round256(4 * sizeof(HeadConsts));       // warning word
static constexpr size_t SS_CHUNKS = SS_STATUS + 256;                                     // second-stage sums of the pushes
static constexpr size_t SS_PARTIALS = SS_CHUNKS + round256(sizeof(double) * REBXAMD_MAX_OPS * 64 * 3);
static_assert(SS_PARTIALS + (1u << 20) <= STATIC_SCRATCH_BYTES, "the fixed scratch array is too small for its tables");

static constexpr int STATIC_MAX_DEVICES = 64;
// One slot per GPU: its own lock, held for the whole host side of a call (set-up, argument checks, launches -- the
// launches are asynchronous, so that is microseconds).  The per-device context below is host state that a call changes
// (warning word, table index, error text); two host threads on one GPU take turns on it.  What the lock cannot order is
// the work on the GPU: calls queued on different streams of one GPU still share the device array.
struct StaticSlot { std::mutex lock; bool ready = false; rebxamd_ctx ctx; };
static StaticSlot g_static[STATIC_MAX_DEVICES];

static void copy_msg(char* dst, int len, const char* msg) {
    if (!dst || len <= 0) return;
    snprintf(dst, (size_t)len, "%s", msg);
}

extern "C" int rebxamd_apply_static(const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode, uint32_t* warnings_dev, void* stream,
                                    char* errbuf, int errbuf_len) {
    copy_msg(errbuf, errbuf_len, "");
    if (!view || nops < 0 || nops > REBXAMD_MAX_OPS || (nops > 0 && !ops) || view->n < 0 || view->n_head < 0 ||
        (mode != REBXAMD_MODE_FAST && mode != REBXAMD_MODE_EXACT)) {
        copy_msg(errbuf, errbuf_len, "rebxamd_apply_static: bad argument (at most REBXAMD_MAX_OPS ops)");
        return REBXAMD_ERR_ARG;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= STATIC_MAX_DEVICES) {
        copy_msg(errbuf, errbuf_len, "rebxamd_apply_static: no current device");
        return REBXAMD_ERR_HIP;
    }
    StaticSlot& slot = g_static[dev];
    std::lock_guard<std::mutex> lock(slot.lock);
    {
        if (!slot.ready) {
            // set-up, once per device, with that device current
            unsigned char* ptr = nullptr;
            const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || ptr == nullptr) {
                copy_msg(errbuf, errbuf_len, "rebxamd_apply_static: hipGetSymbolAddress(scratch) failed");
                return REBXAMD_ERR_HIP;
            }
            rebxamd_ctx& c = slot.ctx;
            c.device = dev;
            c.fixed_scratch = true;
            c.hc = reinterpret_cast<HeadConsts*>(ptr + SS_TABLES);
            c.status = reinterpret_cast<unsigned*>(ptr + SS_STATUS);
            c.chunk_sums = reinterpret_cast<double*>(ptr + SS_CHUNKS);
            c.partials = reinterpret_cast<double*>(ptr + SS_PARTIALS);
            c.partials_cap = STATIC_SCRATCH_BYTES - SS_PARTIALS;
            slot.ready = true;
        }
    }
    rebxamd_ctx* const ctx = &slot.ctx;
    hipStream_t st = (hipStream_t)stream;
    // Refused up front: the forces whose workspace grows with the number of particles.
    const long long K = view->n_head < view->n ? view->n_head : view->n;
    const bool short_head = (view->tail_massless != 0 || K == view->n) && K >= 1 && K <= REBXAMD_MAX_HEAD;
    for (int k = 0; k < nops; k++) {
        const bool com = ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE;
        if (ops[k].kind == REBXAMD_OP_GR_FULL || ((ops[k].kind == REBXAMD_OP_GR || com) && !short_head)) {
            copy_msg(errbuf, errbuf_len, "rebxamd_apply_static: gr_full, and gr or Jacobi/barycentric modify_orbits_forces with massive bodies beyond a "
                                         "64-body prefix, need workspace in proportion to the particles; use rebxamd_apply");
            return REBXAMD_ERR_UNSUPPORTED;
        }
    }
    unsigned* const own_status = ctx->status;
    if (warnings_dev) ctx->status = warnings_dev;           // kernels raise their warning bits in the caller's word
    else if (hipMemsetAsync(own_status, 0, sizeof(unsigned), st) != hipSuccess) {
        copy_msg(errbuf, errbuf_len, "rebxamd_apply_static: hipMemsetAsync failed");
        return REBXAMD_ERR_HIP;
    }
    ctx->hc_stage = 0;
    const int rc = apply_ops(ctx, view, ops, nops, mode, st);
    ctx->status = own_status;
    if (rc != REBXAMD_OK) {
        std::string msg = "rebxamd_apply_static: " + ctx->err;
        if (rc == REBXAMD_ERR_UNSUPPORTED && ctx->err.find("fixed scratch") != std::string::npos)
            msg += " (limit: pushing ops x ceil(n / 512) x 24 bytes <= " + std::to_string(STATIC_SCRATCH_BYTES - SS_PARTIALS) + ")";
        copy_msg(errbuf, errbuf_len, msg.c_str());
    }
    return rc;
}

extern "C" int rebxamd_apply_targets(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode,
                                     int64_t target_lo, double* sums, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    REFUSE_DURING_RUN("rebxamd_apply_targets");
    if (!view || nops <= 0 || nops > REBXAMD_MAX_OPS || !ops || !sums || view->n < 0 || target_lo < 0)
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_apply_targets: bad argument");
    for (int k = 0; k < nops; k++)
        if (ops[k].kind == REBXAMD_OP_GR_FULL) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_apply_targets: gr_full is sharded by rows, not by particles");
    HIP_TRY(hipSetDevice(ctx->device));
    return run_streaming(ctx, view, ops, 0, nops, mode, (hipStream_t)stream, target_lo, sums, false);
}

extern "C" int rebxamd_add_pushes(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, const double* sums, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    REFUSE_DURING_RUN("rebxamd_add_pushes");
    if (!view || nops <= 0 || nops > REBXAMD_MAX_OPS || !ops || !sums) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_add_pushes: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return run_streaming(ctx, view, ops, 0, nops, REBXAMD_MODE_FAST, (hipStream_t)stream, 0, const_cast<double*>(sums), true);
}

// ------------------------------------------------------------------ force integrators, kick, drift

// whether rebxamd_apply would run the whole op list as a single streaming run
static bool one_streaming_run(const rebxamd_op* ops, int nops) {
    if (nops <= 0 || nops > REBXAMD_MAX_OPS) return false;
    int ngr = 0, ncom = 0;
    for (int k = 0; k < nops; k++) {
        if (ops[k].kind == REBXAMD_OP_GR_FULL) return false;
        if (ops[k].kind == REBXAMD_OP_GR && ++ngr > 1) return false;
        if (ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE && ++ncom > 1) return false;
    }
    return true;
}

static int check_step_view(rebxamd_ctx* ctx, const rebxamd_view* v, const char* who, bool need_pos, bool need_acc) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!v || v->n < 0) return fail(ctx, REBXAMD_ERR_ARG, who);
    if (v->n == 0) return REBXAMD_OK;
    if (!v->vx || !v->vy || !v->vz) return fail(ctx, REBXAMD_ERR_ARG, "velocity columns missing");
    if (need_pos && (!v->x || !v->y || !v->z)) return fail(ctx, REBXAMD_ERR_ARG, "position columns missing");
    if (need_acc && (!v->ax || !v->ay || !v->az)) return fail(ctx, REBXAMD_ERR_ARG, "acceleration columns missing");
    return REBXAMD_OK;
}

extern "C" int rebxamd_kick(rebxamd_ctx* ctx, const rebxamd_view* view, double dt, int mode, void* stream) {
    const int rc = check_step_view(ctx, view, "rebxamd_kick: bad argument", false, true);
    if (rc != REBXAMD_OK || view->n == 0) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    double* const v[3] = {const_cast<double*>(view->vx), const_cast<double*>(view->vy), const_cast<double*>(view->vz)};
    const double* const cv[3] = {view->vx, view->vy, view->vz};
    const double* const a[3] = {view->ax, view->ay, view->az};
    HIP_TRY(launch_lincomb(v, cv, a, dt, view->n, mode == REBXAMD_MODE_EXACT, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_drift(rebxamd_ctx* ctx, const rebxamd_view* view, double dt, int mode, void* stream) {
    const int rc = check_step_view(ctx, view, "rebxamd_drift: bad argument", true, false);
    if (rc != REBXAMD_OK || view->n == 0) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    double* const x[3] = {const_cast<double*>(view->x), const_cast<double*>(view->y), const_cast<double*>(view->z)};
    const double* const cx[3] = {view->x, view->y, view->z};
    const double* const v[3] = {view->vx, view->vy, view->vz};
    HIP_TRY(launch_lincomb(x, cx, v, dt, view->n, mode == REBXAMD_MODE_EXACT, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_integrate_force(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int integrator,
                                       double dt, int mode, int* iterations, void* stream) {
    if (ctx) REFUSE_DURING_RUN("rebxamd_integrate_force");
    if (iterations) *iterations = 0;
    int rc = check_step_view(ctx, view, "rebxamd_integrate_force: bad argument", true, true);
    if (rc != REBXAMD_OK) return rc;
    if (nops < 0 || (nops > 0 && !ops)) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_integrate_force: bad op list");
    if (mode != REBXAMD_MODE_FAST && mode != REBXAMD_MODE_EXACT) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_integrate_force: bad mode");
    hipStream_t st = (hipStream_t)stream;
    const bool exact = mode == REBXAMD_MODE_EXACT;
    const long long n = view->n;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool known = integrator == REBXAMD_INTEGRATOR_EULER || integrator == REBXAMD_INTEGRATOR_RK2 || integrator == REBXAMD_INTEGRATOR_RK4 ||
                       integrator == REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT;
    // Massive bodies in a short prefix, massless bodies behind it, the stack one streaming run: the
    // prefix goes through the separate passes below (on a view of the prefix alone, one head table
    // per stage), and ONE pass over the massless bodies evaluates every stage of their update in
    // registers: they are read once and only their velocities and first-stage accelerations written.
    const long long K = view->n_head < n ? view->n_head : n;
    bool carried = known && ctx->carried_steps && one_streaming_run(ops, nops) &&
                   view->tail_massless != 0 && K >= 1 && K <= REBXAMD_MAX_HEAD && K < n && view->m != nullptr;
    for (int k = 0; k < nops && carried; k++) {
        const bool com = ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE;
        const int src = (ops[k].kind == REBXAMD_OP_GR || ops[k].kind == REBXAMD_OP_GR_POTENTIAL) ? 0 : ops[k].src;
        if (!com && (src < 0 || src >= K)) carried = false;
    }
    // At most 64 bodies, all of them head (also the prefix of a carried update, which comes back here on a view of its own):
    // every stage's pass and every update in one launch of one wavefront -- the same functions in the same order as below.
    if (!carried && known && ctx->head_one_launch && n >= 1 && n <= REBXAMD_MAX_HEAD &&
        (view->tail_massless != 0 || K == n) && K == n && one_streaming_run(ops, nops) && view->ax && view->ay && view->az) {
        const View hv = make_view(view);
        Program prog;
        int max_src = -1;
        rc = build_program(ctx, view, hv, ops, 0, nops, prog, max_src);
        if (rc != REBXAMD_OK) return rc;
        if (max_src < n) {
            const size_t stride = ((size_t)n + 1) & ~(size_t)1;
            rc = ensure(ctx, &ctx->integ, &ctx->integ_cap, (9 * stride + 1) * sizeof(double), st);
            if (rc != REBXAMD_OK) return rc;
            if (integrator == REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT) {
                int* const rounds = reinterpret_cast<int*>(ctx->integ + 9 * stride);
                HIP_TRY(launch_head_implicit_midpoint(hv, prog, ctx->hc, ctx->status, dt, ctx->integ, (long long)stride, rounds, exact, st));
                ctx->hc_stage = 0;
                if (iterations) {
                    int h = 0;
                    { const int rb = read_back(ctx, &h, rounds, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
                    *iterations = h;
                }
                return REBXAMD_OK;
            }
            double c0 = dt, c1 = 0., c2 = 0.;
            if (integrator == REBXAMD_INTEGRATOR_RK2) { c0 = 2. * dt / 3.; c1 = dt / 4.; c2 = 3. * dt / 4.; }
            if (integrator == REBXAMD_INTEGRATOR_RK4) { c0 = dt / 2.; c1 = dt; c2 = dt / 6.; }
            HIP_TRY(launch_head_integrate(hv, prog, ctx->hc, ctx->status, integrator, c0, c1, c2, ctx->integ, (long long)stride, exact, st));
            ctx->hc_stage = 0;
            return REBXAMD_OK;
        }
    }
    if (!carried) {
        // integrate_force.c:46 -- the simulation's accelerations are cleared first, whatever the scheme
        // (the carried pass of an explicit scheme overwrites those of the massless bodies, and the prefix
        // clears its own; the carried implicit scheme clears them in its first pass and leaves them cleared)
        rc = rebxamd_reset_accelerations(ctx, view, stream);
        if (rc != REBXAMD_OK || n == 0) return rc;
    }
    if (!known) return REBXAMD_OK;               // REBX_INTEGRATOR_NONE and unknown values do nothing (integrate_force.c:69-76)

    {
        if (carried && integrator == REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT) {
            // Implicit midpoint: per round, the prefix takes the separate passes (force pass at the
            // averaged state, update + averaged state for the next round) and ONE pass over the massless
            // bodies forms their averaged velocities, evaluates the stack and updates their estimate.
            const int nb = stream_block_count(K, n, true);
            const size_t stride = ((size_t)n + 1) & ~(size_t)1;
            const size_t hs = ((size_t)K + 1) & ~(size_t)1;
            rc = ensure(ctx, &ctx->integ, &ctx->integ_cap, (3 * stride + 6 * hs + 2 * (size_t)(nb + 1) + 2 + 1) * sizeof(double), st);
            if (rc != REBXAMD_OK) return rc;
            double* w = ctx->integ;
            // The velocity columns themselves hold the estimate of the final velocities (it starts as the
            // velocities at the start of the step and ends as the result: no copy back); the velocities at
            // the start of the step are kept in the workspace and read from there.
            double* const orig[3] = {w, w + stride, w + 2 * stride};
            double* const hv[3] = {w + 3 * stride, w + 3 * stride + hs, w + 3 * stride + 2 * hs};          // prefix: averaged velocities
            double* const ha[3] = {w + 3 * stride + 3 * hs, w + 3 * stride + 4 * hs, w + 3 * stride + 5 * hs};   // prefix: accelerations there
            double* partials = w + 3 * stride + 6 * hs;                              // pair 0: the prefix; pairs 1..nb: the pass behind it
            double* sums = partials + 2 * (size_t)(nb + 1);
            unsigned* stop = reinterpret_cast<unsigned*>(sums + 2);                // [0] converged, [1] rounds counted
            const double* const cv3[3] = {view->vx, view->vy, view->vz};
            double* const v3[3] = {const_cast<double*>(view->vx), const_cast<double*>(view->vy), const_cast<double*>(view->vz)};
            if (!view->ax || !view->ay || !view->az) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_integrate_force: acceleration columns missing");
            double* const a3[3] = {view->ax, view->ay, view->az};
            HIP_TRY(launch_im_begin(orig, cv3, a3, n, st));          // keeps the starting velocities, clears the accelerations
            for (int k = 0; k < 3; k++) HIP_TRY(hipMemcpyAsync(hv[k], cv3[k], (size_t)K * sizeof(double), hipMemcpyDeviceToDevice, st));
            const double* const corig[3] = {orig[0], orig[1], orig[2]};
            rebxamd_view start = *view;                 // the bodies with their velocities at the start of the step
            start.vx = orig[0]; start.vy = orig[1]; start.vz = orig[2];
            HIP_TRY(hipMemsetAsync(ha[0], 0, 3 * hs * sizeof(double), st));
            rebxamd_view head = *view;
            head.n = K; head.n_head = K;
            head.vx = hv[0]; head.vy = hv[1]; head.vz = hv[2]; head.ax = ha[0]; head.ay = ha[1]; head.az = ha[2];
            StepArgs sa = {0., 0., nullptr, EPI_IM, dt, 0., 0., {v3[0], v3[1], v3[2]}, partials + 2, nullptr, 0., {}};
            sa.stop = stop;
            HIP_TRY(hipMemsetAsync(stop, 0, 2 * sizeof(unsigned), st));
            // Rounds are queued ahead of the host: the sums are tested on the device, and once the test is
            // met the kernels of the rounds still in the queue return at once.  The host looks after four
            // rounds (most steps need three), then after every second one, up to the reference's ten.
            int it = 10, queued = 0;
            while (queued < 10) {
                const int batch = queued == 0 ? 4 : 2;
                for (int b = 0; b < batch && queued < 10; b++, queued++) {
                    ctx->hc_stage = 0;
                    rc = rebxamd_apply(ctx, &head, ops, nops, mode, stream);                 // prefix forces + head table at the averaged state
                    if (rc != REBXAMD_OK) return rc;
                    rc = run_streaming(ctx, &start, ops, 0, nops, mode, st, -1, nullptr, false, &sa);
                    if (rc != REBXAMD_OK) return rc == STEP_NOT_CARRIED ? fail(ctx, REBXAMD_ERR_ARG, "rebxamd_integrate_force: layout changed between passes") : rc;
                    HIP_TRY(launch_im_update(v3, corig, hv, ha, dt, K, partials, nullptr, exact, st, stop));
                    HIP_TRY(launch_im_sum(partials, nb + 1, sums, st, stop));
                }
                unsigned h[2] = {0u, 0u};
                { const int rb = read_back(ctx, h, stop, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
                if (h[0] != 0u) { it = (int)h[1] - 1; break; }            // the index of the round that met the test (compare(), :57)
            }
            if (it == 10) {
                unsigned cur = 0;
                { const int rb = read_back(ctx, &cur, ctx->status, sizeof(cur), st); if (rb != REBXAMD_OK) return rb; }
                cur |= WARN_IM_ITER;
                HIP_TRY(hipMemcpyAsync(ctx->status, &cur, sizeof(cur), hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            if (iterations) *iterations = it;
            return REBXAMD_OK;
        }
        if (carried) {
            rebxamd_view head = *view;
            head.n = K; head.n_head = K;
            const bool saved = ctx->carried_steps;
            ctx->carried_steps = false;                        // the prefix itself takes the separate passes
            rc = rebxamd_integrate_force(ctx, &head, ops, nops, integrator, dt, mode, nullptr, stream);
            ctx->carried_steps = saved;
            ctx->hc_stage = 0;
            if (rc != REBXAMD_OK) return rc;
            StepArgs sa = {0., 0., nullptr, EPI_EULER, dt, 0., 0., {nullptr, nullptr, nullptr}, nullptr, nullptr, 0.};
            if (integrator == REBXAMD_INTEGRATOR_RK2) { sa.epi = EPI_RK2; sa.c0 = 2. * dt / 3.; sa.c1 = dt / 4.; sa.c2 = 3. * dt / 4.; }
            if (integrator == REBXAMD_INTEGRATOR_RK4) { sa.epi = EPI_RK4; sa.c0 = dt / 2.; sa.c1 = dt; sa.c2 = dt / 6.; }
            rc = run_streaming(ctx, view, ops, 0, nops, mode, st, -1, nullptr, false, &sa);
            if (rc != STEP_NOT_CARRIED) return rc;
            return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_integrate_force: layout changed between passes");
        }
    }

    double* const v[3] = {const_cast<double*>(view->vx), const_cast<double*>(view->vy), const_cast<double*>(view->vz)};
    const double* const cv[3] = {view->vx, view->vy, view->vz};
    const double* const ca1[3] = {view->ax, view->ay, view->az};

    if (integrator == REBXAMD_INTEGRATOR_EULER) {              // integrator_euler.c:33-41
        rc = rebxamd_apply(ctx, view, ops, nops, mode, stream);
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_lincomb(v, cv, ca1, dt, n, exact, st));
        return REBXAMD_OK;
    }

    const long long npart = im_partial_count(n);
    const size_t stride = ((size_t)n + 1) & ~(size_t)1;         // keeps every column 16-byte aligned
    rc = ensure(ctx, &ctx->integ, &ctx->integ_cap, (9 * stride + 2 * (size_t)npart + 2) * sizeof(double), st);
    if (rc != REBXAMD_OK) return rc;
    double* w = ctx->integ;
    double* const sv[3] = {w, w + stride, w + 2 * stride};                       // stage velocities
    double* const ka[3] = {w + 3 * stride, w + 4 * stride, w + 5 * stride};     // stage accelerations
    double* const kb[3] = {w + 6 * stride, w + 7 * stride, w + 8 * stride};
    const double* const cka[3] = {ka[0], ka[1], ka[2]};
    const double* const ckb[3] = {kb[0], kb[1], kb[2]};
    double* partials = w + 9 * stride;
    double* sums = partials + 2 * npart;
    rebxamd_view sa = *view, sb = *view;                        // stage states share positions, masses, parameters
    sa.vx = sv[0]; sa.vy = sv[1]; sa.vz = sv[2]; sa.ax = ka[0]; sa.ay = ka[1]; sa.az = ka[2];
    sb.vx = sv[0]; sb.vy = sv[1]; sb.vz = sv[2]; sb.ax = kb[0]; sb.ay = kb[1]; sb.az = kb[2];
    // the reference copies the particle array after the reset: stage accelerations start at zero
    HIP_TRY(hipMemsetAsync(ka[0], 0, 6 * stride * sizeof(double), st));

    // each stage's head tables are kept (ctx->hc[stage]) for a pass that evaluates all stages at once
    struct StageReset { rebxamd_ctx* c; ~StageReset() { c->hc_stage = 0; } } stage_reset{ctx};
    if (integrator == REBXAMD_INTEGRATOR_RK2) {                // integrator_rk2.c:38-67
        rc = rebxamd_apply(ctx, view, ops, nops, mode, stream);
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_lincomb(sv, cv, ca1, 2. * dt / 3., n, exact, st));
        ctx->hc_stage = 1;
        rc = rebxamd_apply(ctx, &sa, ops, nops, mode, stream);
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_rk2_final(v, ca1, cka, dt / 4., 3. * dt / 4., n, exact, st));
        return REBXAMD_OK;
    }
    if (integrator == REBXAMD_INTEGRATOR_RK4) {                // integrator_rk4.c:40-91
        const double dt2 = dt / 2.;
        rc = rebxamd_apply(ctx, view, ops, nops, mode, stream);                 // k1
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_lincomb(sv, cv, ca1, dt2, n, exact, st));
        ctx->hc_stage = 1;
        rc = rebxamd_apply(ctx, &sa, ops, nops, mode, stream);                  // k2 -> ka
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_lincomb(sv, cv, cka, dt2, n, exact, st));
        ctx->hc_stage = 2;
        rc = rebxamd_apply(ctx, &sb, ops, nops, mode, stream);                  // k3 -> kb
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_rk4_mid(sv, cv, ka, kb, dt, n, exact, st));              // kb = k3 + k2, ka = 0
        ctx->hc_stage = 3;
        rc = rebxamd_apply(ctx, &sa, ops, nops, mode, stream);                  // k4 -> ka
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_rk4_final(v, ca1, cka, ckb, dt / 6., n, exact, st));
        return REBXAMD_OK;
    }
    // implicit midpoint (integrator_implicit_midpoint.c:83-124): sv = averaged velocities,
    // ka = accelerations at the averaged state, kb = current estimate of the final velocities
    for (int k = 0; k < 3; k++) {
        HIP_TRY(hipMemcpyAsync(sv[k], cv[k], (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(kb[k], cv[k], (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    int it = 0;
    for (it = 0; it < 10; it++) {
        rc = rebxamd_apply(ctx, &sa, ops, nops, mode, stream);
        if (rc != REBXAMD_OK) return rc;
        HIP_TRY(launch_im_update(kb, cv, sv, ka, dt, n, partials, sums, exact, st));
        double h[2] = {0., 0.};
        { const int rb = read_back(ctx, h, sums, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
        if (h[0] / h[1] < 2.220446049250313e-16 * 2.220446049250313e-16) break;   // compare(), :57
    }
    if (it == 10) {
        unsigned cur = 0;
        { const int rb = read_back(ctx, &cur, ctx->status, sizeof(cur), st); if (rb != REBXAMD_OK) return rb; }
        cur |= WARN_IM_ITER;
        HIP_TRY(hipMemcpyAsync(ctx->status, &cur, sizeof(cur), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (int k = 0; k < 3; k++) HIP_TRY(hipMemcpyAsync(v[k], ckb[k], (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (iterations) *iterations = it;
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ Wisdom-Holman steps, one dominant mass + massless bodies

extern "C" int rebxamd_wh_drift(rebxamd_ctx* ctx, const rebxamd_view* view, double dt, int mode, void* stream) {
    if (ctx) REFUSE_DURING_RUN("rebxamd_wh_drift");
    const int rc = check_step_view(ctx, view, "rebxamd_wh_drift: bad argument", true, false);
    if (rc != REBXAMD_OK || view->n == 0) return rc;
    if (!view->m) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_drift: mass column missing");
    if (!(view->n_head <= 1 && view->tail_massless)) return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "rebxamd_wh_drift handles one massive body at index 0 followed by massless ones (n_head = 1, tail_massless = 1)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    double* const x[3] = {const_cast<double*>(view->x), const_cast<double*>(view->y), const_cast<double*>(view->z)};
    double* const v[3] = {const_cast<double*>(view->vx), const_cast<double*>(view->vy), const_cast<double*>(view->vz)};
    HIP_TRY(launch_wh_drift(x, v, nullptr, view->m, view->G, dt, 0., view->n, ctx->star, mode == REBXAMD_MODE_EXACT, st));
    return REBXAMD_OK;
}

constexpr int kMaxWhBatch = 64;

// tables and records of the steps of a batch
static int ensure_wh_batch(rebxamd_ctx* ctx) {
    // (two copies: the pass of one batch fills the tables of the next)
    if (!ctx->hc_batch) HIP_TRY(hipMalloc((void**)&ctx->hc_batch, 2 * (size_t)kMaxWhBatch * sizeof(HeadConsts)));
    if (!ctx->star_batch) HIP_TRY(hipMalloc((void**)&ctx->star_batch, 2 * (size_t)kMaxWhBatch * 16 * sizeof(double)));
    return REBXAMD_OK;
}

extern "C" int rebxamd_wh_batch(rebxamd_ctx* ctx, int steps_per_pass) {
    if (!ctx) return REBXAMD_ERR_ARG;
    REFUSE_DURING_RUN("rebxamd_wh_batch");
    if (steps_per_pass < 1 || steps_per_pass > kMaxWhBatch) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_batch: between 1 and 64 steps per pass");
    ctx->wh_batch = steps_per_pass;
    return REBXAMD_OK;
}

extern "C" int rebxamd_wh_steps(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, double dt, int64_t nsteps,
                                int mode, void* stream) {
    if (ctx) REFUSE_DURING_RUN("rebxamd_wh_steps");
    int rc = check_step_view(ctx, view, "rebxamd_wh_steps: bad argument", true, true);
    if (rc != REBXAMD_OK) return rc;
    if (nsteps < 0 || nops < 0 || (nops > 0 && !ops)) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: bad argument");
    if (view->n == 0 || nsteps == 0) return REBXAMD_OK;
    if (!view->m) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: mass column missing");
    const long long K = view->n_head < view->n ? view->n_head : view->n;
    if (!(K <= REBXAMD_MAX_HEAD && (view->tail_massless || K == view->n)))
        return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "rebxamd_wh_steps handles a prefix of at most 64 massive bodies (star, planets) followed by massless ones (n_head <= 64, tail_massless = 1)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const bool exact = mode == REBXAMD_MODE_EXACT;
    if (K > 1) {
        // Star + planets + massless bodies.  The head steps in Jacobi coordinates in one wavefront; the pass over the massless
        // bodies carries forces, the pull of the head bodies, kick and drift about the head's centre of mass.
        if (nops > 0 && !one_streaming_run(ops, nops))
            return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "rebxamd_wh_steps with several massive bodies: the force stack must be one streaming run (no gr_full, one gr, one Jacobi/barycentric modify_orbits_forces)");
        for (int k = 0; k < nops; k++) {
            const bool com = ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE;
            const int src = (ops[k].kind == REBXAMD_OP_GR || ops[k].kind == REBXAMD_OP_GR_POTENTIAL) ? 0 : ops[k].src;
            if (!com && (src < 0 || src >= K)) return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "rebxamd_wh_steps: every op's source must be one of the massive bodies");
        }
        if (!view->x || !view->ax || !view->ay || !view->az) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: position/acceleration columns missing");
        double* const lead = ctx->star + 32;       // record of the head's centre of mass at the start of the run, filled by the first head launch
        struct StageReset { rebxamd_ctx* c; ~StageReset() { c->hc_stage = 0; } } stage_reset{ctx};
        // With massless bodies behind the head, the pass of step k also runs, in one extra workgroup, the head pass and the head
        // step of step k + 1 (they never read a massless body) into the other copy of the tables and of the record: one launch
        // per step after the first.
        const bool chain = ctx->carried_steps && view->n > K;
        // The massless bodies feel the head but not one another: once the head has gone through a batch of steps (one
        // wavefront, tables and frame record of every step kept), one pass takes every massless body through the same steps
        // in registers, as behind a single massive body.
        const int batch = (chain && ctx->wh_batch > 1 && !ctx->fixed_scratch) ? ctx->wh_batch : 1;
        if (batch > 1 && nsteps > 1) {
            rc = ensure_wh_batch(ctx);
            if (rc != REBXAMD_OK) return rc;
            bool ready = false;             // this batch's tables were filled by the pass of the batch before
            for (int64_t k = 0, bi = 0; k < nsteps; bi++) {
                const int nsub = (int)(nsteps - k < batch ? nsteps - k : batch);
                const bool ends = k + nsub == nsteps;
                const int cur = (int)(bi & 1), nxt = cur ^ 1;
                StepArgs sa = {ends ? 0.5 * dt : dt, dt, ctx->star_batch + (size_t)cur * kMaxWhBatch * 16, EPI_WHB, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr,
                               k == 0 ? lead : nullptr, 0.5 * dt, {}};
                sa.wh_head = (int)K;
                sa.jac = ctx->whead;
                sa.lead_in_head = k == 0 ? 1 : 0;
                sa.nsub = nsub; sa.h_sub = dt; sa.tabs = ctx->hc_batch + (size_t)cur * kMaxWhBatch;
                sa.head_ready = ready ? 1 : 0;
                const int64_t left = nsteps - (k + nsub);
                ready = false;
                if (nsub > 1 && left > 1) {     // (a last batch of one step makes its own head pass)
                    sa.next_nsub = (int)(left < batch ? left : batch);
                    sa.next_hc = ctx->hc_batch + (size_t)nxt * kMaxWhBatch; sa.next_star = ctx->star_batch + (size_t)nxt * kMaxWhBatch * 16;
                    sa.next_h = left <= batch ? 0.5 * dt : dt;
                    ready = true;
                }
                if (nsub == 1) {            // a last batch of one step is an ordinary pass behind a head pass and a head step of their own
                    sa.nsub = 0; sa.tabs = nullptr; sa.star = ctx->star; sa.epi = EPI_WH;
                    ctx->hc_stage = 0;
                }
                rc = run_streaming(ctx, view, ops, 0, nops, mode, st, -1, nullptr, false, &sa);
                if (rc == STEP_NOT_CARRIED) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: the force stack cannot ride in one pass over this layout");
                if (rc != REBXAMD_OK) return rc;
                k += nsub;
            }
            return REBXAMD_OK;
        }
        for (int64_t k = 0; k < nsteps; k++) {
            const double h = k + 1 < nsteps ? dt : 0.5 * dt;
            const int cur = chain ? (int)(k & 1) : 0, nxt = cur ^ 1;
            ctx->hc_stage = cur;
            StepArgs sa = {h, dt, ctx->star + 16 * cur, EPI_WH, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr, k == 0 ? lead : nullptr, 0.5 * dt, {}};
            sa.wh_head = (int)K;
            sa.jac = ctx->whead;
            sa.lead_in_head = k == 0 ? 1 : 0;
            if (chain && k + 1 < nsteps) { sa.next_hc = ctx->hc + nxt; sa.next_star = ctx->star + 16 * nxt; sa.next_h = k + 2 < nsteps ? dt : 0.5 * dt; }
            sa.head_ready = (chain && k > 0) ? 1 : 0;
            rc = run_streaming(ctx, view, ops, 0, nops, mode, st, -1, nullptr, false, &sa);
            if (rc == STEP_NOT_CARRIED) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: the force stack cannot ride in one pass over this layout");
            if (rc != REBXAMD_OK) return rc;
        }
        return REBXAMD_OK;
    }
    double* const x[3] = {const_cast<double*>(view->x), const_cast<double*>(view->y), const_cast<double*>(view->z)};
    double* const v[3] = {const_cast<double*>(view->vx), const_cast<double*>(view->vy), const_cast<double*>(view->vz)};
    double* const a[3] = {view->ax, view->ay, view->az};
    // Consecutive half drifts of neighbouring steps are one Kepler drift of a whole step (the
    // flow of the two-body problem composes exactly); only the ends of the run are half steps.
    // When the whole stack is one streaming run whose sources are body 0, the pass over the massless
    // bodies carries the kick and the drift as well (their accelerations never leave registers), and
    // in the first step also the leading half drift: one head pass and one pass per step.
    bool carried = ctx->carried_steps && one_streaming_run(ops, nops) && view->n_head == 1 && view->n > 1;
    for (int k = 0; k < nops && carried; k++) {
        const bool com = ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE;
        const int src = (ops[k].kind == REBXAMD_OP_GR || ops[k].kind == REBXAMD_OP_GR_POTENTIAL) ? 0 : ops[k].src;
        if (!com && src != 0) carried = false;
    }
    if (carried) {
        // (the acceleration columns are neither read nor written on this path: body 0's starts from zero inside the head
        // pass, the others' never leave registers)
        double* const star0 = ctx->star + 32;      // record of body 0 for the leading half drift, filled by the first head pass
        // With a one-body head pass (no barycentric op) the pass of step k also runs the head pass of
        // step k + 1 in one extra workgroup, into the other copy of the tables and of body 0's record:
        // one launch per step after the first.
        bool chain = view->vx != nullptr;
        for (int k = 0; k < nops; k++)
            if (ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg == REBX_COORDINATES_BARYCENTRIC) chain = false;
        if (!chain) {
            // the general head pass reads body 0's acceleration column and has no room for the leading drift
            rc = rebxamd_reset_accelerations(ctx, view, stream);
            if (rc != REBXAMD_OK) return rc;
            HIP_TRY(launch_wh_star(x, v, view->m, view->G, 0.5 * dt, star0, st));   // body 0 to mid-step; the others follow inside the pass
        }
        struct StageReset { rebxamd_ctx* c; ~StageReset() { c->hc_stage = 0; } } stage_reset{ctx};
        // The massless bodies are independent of one another and body 0 of them: one pass can take every body through a batch
        // of steps in registers -- positions and velocities make one round trip per batch instead of one per step -- once the
        // head passes of those steps (tables, records of body 0) have been made, one after the other in one wavefront.
        const int batch = (chain && ctx->wh_batch > 1 && !ctx->fixed_scratch) ? ctx->wh_batch : 1;
        if (batch > 1 && nsteps > 1) {
            rc = ensure_wh_batch(ctx);
            if (rc != REBXAMD_OK) return rc;
            bool ready = false;             // this batch's tables were filled by the pass of the batch before
            for (int64_t k = 0, bi = 0; k < nsteps; bi++) {
                const int nsub = (int)(nsteps - k < batch ? nsteps - k : batch);
                const bool ends = k + nsub == nsteps;
                const int cur = (int)(bi & 1), nxt = cur ^ 1;
                StepArgs sa = {ends ? 0.5 * dt : dt, dt, ctx->star_batch + (size_t)cur * kMaxWhBatch * 16, EPI_WHB, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr,
                               k == 0 ? star0 : nullptr, 0.5 * dt, {}};
                sa.lead_in_head = k == 0 ? 1 : 0;
                sa.nsub = nsub; sa.h_sub = dt; sa.tabs = ctx->hc_batch + (size_t)cur * kMaxWhBatch;
                sa.head_ready = ready ? 1 : 0;
                const int64_t left = nsteps - (k + nsub);
                ready = false;
                if (nsub > 1 && left > 1) {     // (a last batch of one step makes its own head pass)
                    sa.next_nsub = (int)(left < batch ? left : batch);
                    sa.next_hc = ctx->hc_batch + (size_t)nxt * kMaxWhBatch; sa.next_star = ctx->star_batch + (size_t)nxt * kMaxWhBatch * 16;
                    sa.next_h = left <= batch ? 0.5 * dt : dt;
                    ready = true;
                }
                if (nsub == 1) {            // a last batch of one step is an ordinary pass; its head pass goes through the same launcher
                    sa.nsub = 0; sa.tabs = nullptr; sa.star = ctx->star; sa.epi = EPI_WH;
                    ctx->hc_stage = 0;
                }
                rc = run_streaming(ctx, view, ops, 0, nops, mode, st, -1, nullptr, false, &sa);
                if (rc == STEP_NOT_CARRIED) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: the force stack cannot ride in one pass over this layout");
                if (rc != REBXAMD_OK) return rc;
                k += nsub;
            }
            return REBXAMD_OK;
        }
        for (int64_t k = 0; k < nsteps; k++) {
            const double h = k + 1 < nsteps ? dt : 0.5 * dt;
            const int cur = chain ? (int)(k & 1) : 0, nxt = cur ^ 1;
            ctx->hc_stage = cur;
            StepArgs sa = {h, dt, ctx->star + 16 * cur, EPI_WH, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr, k == 0 ? star0 : nullptr, 0.5 * dt, {}};
            if (chain && k + 1 < nsteps) { sa.next_hc = ctx->hc + nxt; sa.next_star = ctx->star + 16 * nxt; sa.next_h = k + 2 < nsteps ? dt : 0.5 * dt; }
            sa.head_ready = (chain && k > 0) ? 1 : 0;
            sa.lead_in_head = (chain && k == 0) ? 1 : 0;
            rc = run_streaming(ctx, view, ops, 0, nops, mode, st, -1, nullptr, false, &sa);
            if (rc == STEP_NOT_CARRIED) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_steps: the force stack cannot ride in one pass over this layout");
            if (rc != REBXAMD_OK) return rc;
        }
        return REBXAMD_OK;
    }
    rc = rebxamd_reset_accelerations(ctx, view, stream);
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_wh_drift(x, v, nullptr, view->m, view->G, 0.5 * dt, 0., view->n, ctx->star, exact, st));
    for (int64_t k = 0; k < nsteps; k++) {
        const double h = k + 1 < nsteps ? dt : 0.5 * dt;
        if (nops > 0) {
            rc = rebxamd_apply(ctx, view, ops, nops, mode, stream);
            if (rc != REBXAMD_OK) return rc;
        }
        HIP_TRY(launch_wh_drift(x, v, a, view->m, view->G, h, dt, view->n, ctx->star, exact, st));
    }
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ Wisdom-Holman run left unsynchronised between calls
// Step k of such a run is ONE pass: { drift dt (dt/2 in step 0) about body 0; forces; kick } -- the trailing half drift of
// a step and the leading one of the next are one Kepler drift, carried at the head of the next pass, and the head pass of
// step k + 1 (tables, record and leading drift of body 0) rides in the extra workgroup of pass k.  Body 0's row therefore
// runs one leading drift ahead of the others while the run is open; its records keep what the end of the run needs.

extern "C" int rebxamd_wh_run_begin(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, double dt, int mode, void* stream) {
    int rc = check_step_view(ctx, view, "rebxamd_wh_run_begin: bad argument", true, true);
    if (rc != REBXAMD_OK) return rc;
    if (ctx->run.open) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_begin: a run is already open on this context (close it with rebxamd_wh_run_end)");
    if (nops < 0 || nops > REBXAMD_MAX_OPS || (nops > 0 && !ops)) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_begin: bad argument");
    if (!view->m || !view->x || !view->vx) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_begin: position/velocity/mass columns missing");
    const long long Kh = view->n_head < view->n ? view->n_head : view->n;
    if (Kh > 1) {
        // star + planets + massless bodies: the head steps one step ahead in the extra workgroup of every pass
        bool fits = ctx->carried_steps && Kh <= REBXAMD_MAX_HEAD && view->tail_massless && view->n > Kh && (nops == 0 || one_streaming_run(ops, nops)) &&
                    view->ax && view->ay && view->az;
        for (int k = 0; k < nops && fits; k++) {
            const bool com = ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE;
            const int src = (ops[k].kind == REBXAMD_OP_GR || ops[k].kind == REBXAMD_OP_GR_POTENTIAL) ? 0 : ops[k].src;
            if (!com && (src < 0 || src >= Kh)) fits = false;
        }
        if (!fits)
            return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "rebxamd_wh_run_begin behind several massive bodies handles a head of at most 64 of them followed by massless ones (tail_massless = 1, acceleration columns present) under a stack that is one streaming run of forces whose sources are head bodies; use rebxamd_wh_steps otherwise");
        ctx->run.open = true; ctx->run.stream = stream; ctx->run.k = 0; ctx->run.dt = dt; ctx->run.mode = mode; ctx->run.K = Kh;
        ctx->run.view = *view;
        ctx->run.ops.assign(ops, ops + nops);
        return REBXAMD_OK;
    }
    bool ok = ctx->carried_steps && view->n_head == 1 && view->tail_massless && view->n > 1 && one_streaming_run(ops, nops);
    for (int k = 0; k < nops && ok; k++) {
        const bool com = ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg != REBX_COORDINATES_PARTICLE;
        const int src = (ops[k].kind == REBXAMD_OP_GR || ops[k].kind == REBXAMD_OP_GR_POTENTIAL) ? 0 : ops[k].src;
        if ((!com && src != 0) || (ops[k].kind == REBXAMD_OP_MOF && ops[k].iarg == REBX_COORDINATES_BARYCENTRIC)) ok = false;
    }
    if (!ok)
        return fail(ctx, REBXAMD_ERR_UNSUPPORTED, "rebxamd_wh_run_begin handles one massive body followed by massless ones (n_head = 1, tail_massless = 1) under a non-empty stack that is one streaming run of forces whose source is body 0 (no gr_full, no barycentric modify_orbits_forces); use rebxamd_wh_steps otherwise");
    ctx->run.open = true; ctx->run.stream = stream; ctx->run.k = 0; ctx->run.dt = dt; ctx->run.mode = mode; ctx->run.K = 1;
    ctx->run.view = *view;
    ctx->run.ops.assign(ops, ops + nops);
    return REBXAMD_OK;
}

// Where an unsynchronised run behind several massive bodies keeps its records (slots of 16 doubles in ctx->star): the frame
// records of three consecutive steps in rotation -- pass k drifts the grains about the record of step k - 1, kicks them with
// that of step k, and its extra workgroup writes that of step k + 1 --, the record of the start of the run, the closing one.
static double* run_record(rebxamd_ctx* ctx, int64_t k) { static const int slot[3] = {0, 16, 48}; return ctx->star + slot[((k % 3) + 3) % 3]; }
static double* run_pre(rebxamd_ctx* ctx, int64_t k) { return ctx->whead + 8 * REBXAMD_MAX_HEAD * (1 + (int)(k & 1)); }

extern "C" int rebxamd_wh_run_steps(rebxamd_ctx* ctx, int64_t nsteps, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!ctx->run.open) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_steps: no open run");
    if (stream != ctx->run.stream) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_steps: a run stays on the stream it was opened on");
    if (nsteps < 0) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_steps: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const double dt = ctx->run.dt;
    struct StageReset { rebxamd_ctx* c; ~StageReset() { c->hc_stage = 0; } } stage_reset{ctx};
    if (ctx->run.K > 1) {
        // Step k is one pass: { the grains drift dt (dt/2 in step 0) about the head's centre of mass; forces and the pull of the
        // head bodies; kick }.  The head is one step ahead: its head pass and its own step (kick, whole drift) for step k + 1 ride
        // in the extra workgroup of pass k; the state it has between kick and drift is kept for the end of the run.
        for (int64_t j = 0; j < nsteps; j++) {
            const int64_t k = ctx->run.k;
            const int cur = (int)(k & 1), nxt = cur ^ 1;
            ctx->hc_stage = cur;
            StepArgs sa = {0., dt, run_record(ctx, k), EPI_WH, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr,
                           k == 0 ? ctx->star + 32 : run_record(ctx, k - 1), k == 0 ? 0.5 * dt : dt, {}};
            sa.wh_head = (int)ctx->run.K;
            sa.jac = ctx->whead;
            sa.head_h = dt; sa.pre = run_pre(ctx, k);
            sa.next_hc = ctx->hc + nxt; sa.next_star = run_record(ctx, k + 1); sa.next_h = dt; sa.next_pre = run_pre(ctx, k + 1);
            sa.head_ready = k > 0 ? 1 : 0;
            sa.lead_in_head = k == 0 ? 1 : 0;
            const int rc = run_streaming(ctx, &ctx->run.view, ctx->run.ops.data(), 0, (int)ctx->run.ops.size(), ctx->run.mode, st, -1, nullptr, false, &sa);
            if (rc == STEP_NOT_CARRIED) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_steps: the force stack cannot ride in one pass over this layout");
            if (rc != REBXAMD_OK) return rc;
            ctx->run.k = k + 1;
        }
        return REBXAMD_OK;
    }
    for (int64_t j = 0; j < nsteps; j++) {
        const int64_t k = ctx->run.k;
        const int cur = (int)(k & 1), nxt = cur ^ 1;
        ctx->hc_stage = cur;
        StepArgs sa = {0., dt, ctx->star + 16 * cur, EPI_WH, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr, ctx->star + 48 + 16 * cur, k == 0 ? 0.5 * dt : dt, {}};
        sa.next_hc = ctx->hc + nxt; sa.next_star = ctx->star + 16 * nxt; sa.next_h = 0.;
        sa.next_lead = ctx->star + 48 + 16 * nxt; sa.next_h_lead = dt;
        sa.head_ready = k > 0 ? 1 : 0;
        sa.lead_in_head = k == 0 ? 1 : 0;
        const int rc = run_streaming(ctx, &ctx->run.view, ctx->run.ops.data(), 0, (int)ctx->run.ops.size(), ctx->run.mode, st, -1, nullptr, false, &sa);
        if (rc == STEP_NOT_CARRIED) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_steps: the force stack cannot ride in one pass over this layout");
        if (rc != REBXAMD_OK) return rc;
        ctx->run.k = k + 1;
    }
    return REBXAMD_OK;
}

extern "C" int rebxamd_wh_run_end(rebxamd_ctx* ctx, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!ctx->run.open) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_end: no open run");
    if (stream != ctx->run.stream) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_wh_run_end: a run stays on the stream it was opened on");
    ctx->run.open = false;
    if (ctx->run.k == 0) return REBXAMD_OK;            // nothing was launched: the columns are as they were
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const rebxamd_view* view = &ctx->run.view;
    double* const x[3] = {const_cast<double*>(view->x), const_cast<double*>(view->y), const_cast<double*>(view->z)};
    double* const v[3] = {const_cast<double*>(view->vx), const_cast<double*>(view->vy), const_cast<double*>(view->vz)};
    if (ctx->run.K > 1) {
        // the head back to its state after the kick of the last step (it has run a whole step ahead since) and over the half
        // drift that closes that step; the grains over the same half drift about the head's centre of mass
        const int64_t last_step = ctx->run.k - 1;
        const bool exact = ctx->run.mode == REBXAMD_MODE_EXACT;
        const double h = 0.5 * ctx->run.dt;
        HIP_TRY(launch_wh_head_close(make_view(view), ctx->run.K, view->G, h, run_pre(ctx, last_step), ctx->whead, run_record(ctx, last_step), ctx->star + 80, exact, st));
        HIP_TRY(launch_wh_drift_about(x, v, ctx->run.K, view->n, h, ctx->star + 80, exact, st));
        return REBXAMD_OK;
    }
    // body 0 back to where the last pass's forces saw it (its row ran ahead by the leading drift of a step that never came),
    // then the closing half drift of everything
    const int last = (int)((ctx->run.k - 1) & 1);
    HIP_TRY(launch_wh_star_restore(x, ctx->star + 16 * last, st));
    HIP_TRY(launch_wh_drift(x, v, nullptr, view->m, view->G, 0.5 * ctx->run.dt, 0., view->n, ctx->star + 80, ctx->run.mode == REBXAMD_MODE_EXACT, st));
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ Jacobi/barycentric modify_orbits_forces on particle shards

static View sub_view(const rebxamd_view* u, long long lo, long long hi) {
    View v = make_view(u);
    auto off = [lo](const double* p) { return p ? p + lo : p; };
    v.n = hi - lo; v.n_head = 0;
    v.x = off(v.x); v.y = off(v.y); v.z = off(v.z); v.vx = off(v.vx); v.vy = off(v.vy); v.vz = off(v.vz); v.m = off(v.m);
    v.ax = v.ax ? v.ax + lo : nullptr; v.ay = v.ay ? v.ay + lo : nullptr; v.az = v.az ? v.az + lo : nullptr;
    v.beta = off(v.beta); v.Rt = off(v.Rt); v.k1 = off(v.k1); v.ta = off(v.ta); v.te = off(v.te); v.ti = off(v.ti);
    return v;
}

static int shard_check(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, const char* who) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!view || lo < 0 || hi < lo || hi > view->n) return fail(ctx, REBXAMD_ERR_ARG, who);
    if (view->n > 0 && (!view->x || !view->y || !view->z || !view->vx || !view->vy || !view->vz || !view->m))
        return fail(ctx, REBXAMD_ERR_ARG, "position/velocity/mass columns missing");
    return REBXAMD_OK;
}

// the range that starts at 0 and the range behind it keep separate halves of the workspace
static int shard_workspace(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, hipStream_t st, double** ws) {
    const size_t half = scan_range_doubles(view->n);
    const int rc = ensure(ctx, &ctx->shard, &ctx->shard_cap, 2 * half * sizeof(double), st);
    if (rc != REBXAMD_OK) return rc;
    *ws = ctx->shard + (lo == 0 ? 0 : half);
    return REBXAMD_OK;
}

extern "C" int rebxamd_shard_sums(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, double* sums7, void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_sums: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (!sums7) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_sums: no output");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_range_sums(sub_view(view, lo, hi), ws, sums7, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_shard_mof_forces(rebxamd_ctx* ctx, const rebxamd_view* view, int coordinates, int64_t lo, int64_t hi,
                                        const double* before7, double* push3, void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_mof_forces: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (coordinates != REBX_COORDINATES_JACOBI && coordinates != REBX_COORDINATES_BARYCENTRIC)
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_mof_forces: Jacobi or barycentric coordinates only");
    if (!push3 || (coordinates == REBX_COORDINATES_BARYCENTRIC && !before7)) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_mof_forces: missing vector");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    DevOp op;
    memset(&op, 0, sizeof(op));
    op.kind = REBXAMD_OP_MOF; op.src = -1; op.iarg = coordinates; op.slot = -1;
    HIP_TRY(launch_range_mof_forces(sub_view(view, lo, hi), op, view->G, coordinates == REBX_COORDINATES_BARYCENTRIC, lo == 0, before7, ws, push3,
                                    (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_shard_mof_apply(rebxamd_ctx* ctx, const rebxamd_view* view, int coordinates, int64_t lo, int64_t hi, const double* after3,
                                       void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_mof_apply: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (coordinates != REBX_COORDINATES_JACOBI && coordinates != REBX_COORDINATES_BARYCENTRIC)
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_mof_apply: Jacobi or barycentric coordinates only");
    if (view->n > 0 && (!view->ax || !view->ay || !view->az)) return fail(ctx, REBXAMD_ERR_ARG, "acceleration columns missing");
    if (coordinates == REBX_COORDINATES_BARYCENTRIC && !after3) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_mof_apply: missing vector");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_range_mof_apply(sub_view(view, lo, hi), coordinates == REBX_COORDINATES_BARYCENTRIC, after3, ws, (hipStream_t)stream));
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ gr on particle shards that hold massive bodies

extern "C" int rebxamd_shard_massive(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, int64_t id0, double* records, double* count,
                                     void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_massive: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (!count || (hi > lo && !records) || id0 < 0) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_massive: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_range_massive(sub_view(view, lo, hi), view->G, (double)id0, ws, records, count, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_shard_gr_newton(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, int64_t id0, const double* records,
                                       int64_t nrecords, double* sums10, int mode, void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_gr_newton: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (!sums10 || nrecords < 0 || (nrecords > 0 && !records) || id0 < 0 || (mode != REBXAMD_MODE_FAST && mode != REBXAMD_MODE_EXACT))
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_gr_newton: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_range_gr_newton(sub_view(view, lo, hi), records, nrecords, (double)id0, mode == REBXAMD_MODE_EXACT, ws, sums10, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_shard_gr_forces(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int max_iterations, int64_t lo, int64_t hi,
                                       const double* before10, double* push3, int mode, void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_gr_forces: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (!push3 || (lo > 0 && !before10) || (mode != REBXAMD_MODE_FAST && mode != REBXAMD_MODE_EXACT))
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_shard_gr_forces: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    DevOp op;
    memset(&op, 0, sizeof(op));
    op.kind = REBXAMD_OP_GR; op.src = 0; op.iarg = max_iterations; op.slot = -1; op.a = c; op.c = c * c; op.d = 1. / (c * c);
    HIP_TRY(launch_range_gr_forces(sub_view(view, lo, hi), op, view->G, lo == 0, before10, view->m, mode == REBXAMD_MODE_EXACT, ws, push3, ctx->status,
                                   (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_shard_gr_apply(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, const double* after3, void* stream) {
    int rc = shard_check(ctx, view, lo, hi, "rebxamd_shard_gr_apply: bad argument");
    if (rc != REBXAMD_OK) return rc;
    if (view->n > 0 && (!view->ax || !view->ay || !view->az)) return fail(ctx, REBXAMD_ERR_ARG, "acceleration columns missing");
    HIP_TRY(hipSetDevice(ctx->device));
    double* ws = nullptr;
    rc = shard_workspace(ctx, view, lo, (hipStream_t)stream, &ws);
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_range_gr_apply(sub_view(view, lo, hi), after3, ws, (hipStream_t)stream));
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ operators

extern "C" int rebxamd_modify_mass(rebxamd_ctx* ctx, const rebxamd_view* view, const double* tau_mass, double dt, void* stream) {
    const int rc = check_step_view(ctx, view, "rebxamd_modify_mass: bad argument", true, false);
    if (rc != REBXAMD_OK || view->n == 0) return rc;
    if (!view->m || !tau_mass) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_modify_mass: mass or tau_mass column missing");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int rc2 = ensure(ctx, &ctx->partials, &ctx->partials_cap, (size_t)operator_partial_doubles(view->n) * sizeof(double), st);
    if (rc2 != REBXAMD_OK) return rc2;
    HIP_TRY(launch_modify_mass(make_view(view), tau_mass, dt, ctx->partials, st));
    return REBXAMD_OK;
}

extern "C" int rebxamd_direct_chain(rebxamd_ctx* ctx, int max_rounds, int* last_rounds) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (max_rounds >= 0) ctx->chain_rounds = max_rounds > 64 ? 64 : max_rounds;
    if (last_rounds) *last_rounds = ctx->chain_last_rounds;
    return REBXAMD_OK;
}

extern "C" int rebxamd_modify_orbits_direct(rebxamd_ctx* ctx, const rebxamd_view* view, int coordinates, int primary, const double* p_param,
                                            const double* tau_omega, const double* tau_Omega, double dt, void* stream) {
    if (ctx) REFUSE_DURING_RUN("rebxamd_modify_orbits_direct");
    const int rc = check_step_view(ctx, view, "rebxamd_modify_orbits_direct: bad argument", true, false);
    if (rc != REBXAMD_OK || view->n == 0) return rc;
    if (!view->m) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_modify_orbits_direct: mass column missing");
    if (coordinates != REBX_COORDINATES_JACOBI && coordinates != REBX_COORDINATES_BARYCENTRIC && coordinates != REBX_COORDINATES_PARTICLE)
        return fail(ctx, REBXAMD_ERR_ARG, "modify_orbits_direct: unknown coordinates");
    const long long K = view->n_head < view->n ? view->n_head : view->n;
    if (coordinates == REBX_COORDINATES_PARTICLE && (primary < 0 || primary >= view->n))
        return fail(ctx, REBXAMD_ERR_PARAM, "modify_orbits_direct: PARTICLE coordinates need a primary among the particles");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!((view->tail_massless != 0 || K == view->n) && K >= 1 && K <= REBXAMD_MAX_HEAD && (coordinates != REBX_COORDINATES_PARTICLE || primary < K))) {
        // massive bodies anywhere: the chain of pushes as a fixed-point iteration over suffix sums, the sequential walk if it does not settle
        hipStream_t st = (hipStream_t)stream;
        const View v = make_view(view);
        const int rc2 = ensure(ctx, &ctx->scan, &ctx->scan_cap, (size_t)direct_chain_doubles(v.n) * sizeof(double), st);
        if (rc2 != REBXAMD_OK) return rc2;
        const double pp = p_param ? *p_param : 0.;
        const int has_p = p_param ? 1 : 0;
        const unsigned* ctl = nullptr;
        HIP_TRY(launch_direct_chain_rounds(v, coordinates, primary, pp, has_p, tau_omega, tau_Omega, dt, view->G, ctx->scan, ctx->chain_rounds, &ctl, st));
        if (coordinates != REBX_COORDINATES_PARTICLE) {
            unsigned state[2] = {0, 0};
            const int rc3 = read_back(ctx, state, ctl, sizeof(state), st);
            if (rc3 != REBXAMD_OK) return rc3;
            ctx->chain_last_rounds = state[0] ? (int)state[1] : -1;
            if (!state[0]) HIP_TRY(launch_direct_chain_sequential(v, coordinates, primary, pp, has_p, tau_omega, tau_Omega, dt, view->G, ctx->scan, st));
        } else ctx->chain_last_rounds = 1;
        HIP_TRY(launch_direct_chain_finish(v, coordinates, primary, pp, has_p, tau_omega, tau_Omega, dt, view->G, ctx->scan, st));
        return REBXAMD_OK;
    }
    HIP_TRY(launch_modify_orbits_direct(make_view(view), K, coordinates, primary, p_param ? *p_param : 0., p_param ? 1 : 0, tau_omega, tau_Omega, dt,
                                        view->G, ctx->star, (hipStream_t)stream));
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ potentials

extern "C" int rebxamd_potentials(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, double* out, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    REFUSE_DURING_RUN("rebxamd_potentials");
    if (!view || view->n < 0 || nops < 0 || nops > REBXAMD_MAX_OPS || (nops > 0 && (!ops || !out)))
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_potentials: bad argument");
    for (int k = 0; k < nops; k++) out[k] = 0.;
    if (nops == 0 || view->n == 0) return REBXAMD_OK;
    const View v = make_view(view);
    if (!v.x || !v.y || !v.z || !v.m) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_potentials: position/mass columns missing");
    Program prog;
    memset(&prog, 0, sizeof(prog));
    prog.nops = nops;
    prog.G = view->G;
    for (int k = 0; k < nops; k++) {
        const rebxamd_op& o = ops[k];
        DevOp& d = prog.ops[k];
        d.kind = o.kind; d.src = o.src; d.iarg = o.iarg; d.slot = -1; d.a = o.a; d.b = o.b; d.c = 0.; d.d = 0.;
        switch (o.kind) {
        case REBXAMD_OP_J2: case REBXAMD_OP_J4: case REBXAMD_OP_CENTRAL: break;
        case REBXAMD_OP_TIDES: prog.need_tides = 1; break;
        case REBXAMD_OP_GR_POTENTIAL: d.src = 0; d.c = o.a * o.a; break;
        default: return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_potentials: this op has no potential (J2, J4, central_force, tides_precession, gr_potential do)");
        }
        if (d.src < 0 || d.src >= v.n) return fail(ctx, REBXAMD_ERR_ARG, "op source index out of range");
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const long long nb = potential_block_count(v.n);
    const int rc = ensure(ctx, &ctx->partials, &ctx->partials_cap, ((size_t)nops * (size_t)nb + (size_t)nops) * sizeof(double), st);
    if (rc != REBXAMD_OK) return rc;
    double* dout = ctx->partials + (size_t)nops * (size_t)nb;
    HIP_TRY(launch_potentials(v, prog, ctx->partials, dout, st));
    { const int rb = read_back(ctx, out, dout, (size_t)nops * sizeof(double), st); if (rb != REBXAMD_OK) return rb; }
    return REBXAMD_OK;
}

extern "C" int rebxamd_gr_hamiltonian(rebxamd_ctx* ctx, const rebxamd_view* view, double c, double* H, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!view || view->n < 0 || !H) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_gr_hamiltonian: bad argument");
    *H = 0.;
    if (view->n == 0) return REBXAMD_OK;
    const View v = make_view(view);
    if (!v.x || !v.y || !v.z || !v.vx || !v.vy || !v.vz || !v.m) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_gr_hamiltonian: position/velocity/mass columns missing");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = ensure(ctx, &ctx->scan, &ctx->scan_cap, (scan_workspace_doubles(v.n) + 8) * sizeof(double), st);
    if (rc != REBXAMD_OK) return rc;
    double* dout = ctx->scan + scan_workspace_doubles(v.n);
    HIP_TRY(launch_gr_hamiltonian(v, view->G, c, ctx->scan, dout, st));
    double h[3] = {0., 0., 0.};
    { const int rb = read_back(ctx, h, dout, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
    *H = h[0] + h[1] + h[2] / (c * c);              // T + V_newt + V_PN (gr.c:239-245)
    return REBXAMD_OK;
}

extern "C" int rebxamd_gr_full_hamiltonian(rebxamd_ctx* ctx, const rebxamd_view* view, double c, double* H, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!view || view->n < 0 || !H) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_gr_full_hamiltonian: bad argument");
    *H = 0.;
    if (view->n == 0) return REBXAMD_OK;
    const View v = make_view(view);
    if (!v.x || !v.y || !v.z || !v.vx || !v.vy || !v.vz || !v.m) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_gr_full_hamiltonian: position/velocity/mass columns missing");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = ensure(ctx, &ctx->gf, &ctx->gf_cap, grfull_hamiltonian_doubles(v.n) * sizeof(double), st);
    if (rc != REBXAMD_OK) return rc;
    double* dout = nullptr;
    HIP_TRY(launch_grfull_hamiltonian(v, view->G, c, ctx->gf, &dout, st));
    double h[3] = {0., 0., 0.};
    { const int rb = read_back(ctx, h, dout, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
    *H = h[0] + h[1] + h[2];
    return REBXAMD_OK;
}

// ------------------------------------------------------------------ gr_full

static int grfull_check(rebxamd_ctx* ctx, const rebxamd_view* u, int64_t row0, int64_t row1) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!u || u->n < 0 || row0 < 0 || row1 < row0 || row1 > u->n) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: bad row range");
    if (u->n > 0 && (!u->x || !u->y || !u->z || !u->vx || !u->vy || !u->vz || !u->m)) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: missing column");
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_pot(rebxamd_ctx* ctx, const rebxamd_view* u, double c, int64_t row0, int64_t row1, double* pot, int mode, void* stream) {
    int rc = grfull_check(ctx, u, row0, row1);
    if (rc != REBXAMD_OK) return rc;
    if (!pot) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL output");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool exact = mode == REBXAMD_MODE_EXACT;
    if (!exact) {
        rc = ensure(ctx, &ctx->partials, &ctx->partials_cap, grfull_partial_bytes(u->n, row1 - row0), st);
        if (rc != REBXAMD_OK) return rc;
    }
    HIP_TRY(launch_grfull_pass(0, make_view(u), u->G, c, row0, row1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr, pot, nullptr, nullptr, nullptr, ctx->partials, exact, st));
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_const(rebxamd_ctx* ctx, const rebxamd_view* u, double c, int64_t row0, int64_t row1, const double* pot,
                                    double* acx, double* acy, double* acz, int mode, void* stream) {
    int rc = grfull_check(ctx, u, row0, row1);
    if (rc != REBXAMD_OK) return rc;
    if (!pot || !acx || !acy || !acz) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool exact = mode == REBXAMD_MODE_EXACT;
    if (!exact) {
        rc = ensure(ctx, &ctx->partials, &ctx->partials_cap, grfull_partial_bytes(u->n, row1 - row0), st);
        if (rc != REBXAMD_OK) return rc;
    }
    HIP_TRY(launch_grfull_pass(1, make_view(u), u->G, c, row0, row1, pot, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr, acx, acy, acz, nullptr, ctx->partials, exact, st));
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_sweep(rebxamd_ctx* ctx, const rebxamd_view* u, double c, int64_t row0, int64_t row1,
                                    const double* anx, const double* any_, const double* anz,
                                    const double* aox, const double* aoy, const double* aoz,
                                    const double* acx, const double* acy, const double* acz,
                                    double* outx, double* outy, double* outz, uint32_t* dev_flag, int mode, void* stream) {
    int rc = grfull_check(ctx, u, row0, row1);
    if (rc != REBXAMD_OK) return rc;
    if (!anx || !any_ || !anz || !aox || !aoy || !aoz || !acx || !acy || !acz || !outx || !outy || !outz || !dev_flag)
        return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool exact = mode == REBXAMD_MODE_EXACT;
    if (!exact) {
        rc = ensure(ctx, &ctx->partials, &ctx->partials_cap, grfull_partial_bytes(u->n, row1 - row0), st);
        if (rc != REBXAMD_OK) return rc;
    }
    HIP_TRY(launch_grfull_pass(2, make_view(u), u->G, c, row0, row1, nullptr, anx, any_, anz, aox, aoy, aoz, acx, acy, acz,
                               outx, outy, outz, dev_flag, ctx->partials, exact, st, ctx->gf_stop));
    return REBXAMD_OK;
}

// The potential sums of rows [row0,row1) together with the pair sums of the FIRST sweep (which starts from a_old = 0 and
// therefore needs neither the potentials nor a_const): one pass over the pairs instead of two.  The sweep's sums stay in
// the context until rebxamd_grfull_sweep0_finish adds a_const to them.  Reference-order arithmetic: the potential pass
// alone; the finish call then runs the sweep itself.
extern "C" int rebxamd_grfull_pot_sweep0(rebxamd_ctx* ctx, const rebxamd_view* u, double c, int64_t row0, int64_t row1,
                                         const double* anx, const double* any_, const double* anz, double* pot, int mode, void* stream) {
    if (mode == REBXAMD_MODE_EXACT) return rebxamd_grfull_pot(ctx, u, c, row0, row1, pot, mode, stream);
    int rc = grfull_check(ctx, u, row0, row1);
    if (rc != REBXAMD_OK) return rc;
    if (!pot || !anx || !any_ || !anz) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    rc = ensure(ctx, &ctx->gf4, &ctx->gf4_cap, grfull_partial_bytes(u->n, row1 - row0), st);
    if (rc != REBXAMD_OK) return rc;
    const View v = make_view(u);
    HIP_TRY(launch_grfull_sweep0_pot(v, u->G, c, row0, row1, anx, any_, anz, ctx->gf4, st));
    HIP_TRY(launch_grfull_sweep0_pot_combine(false, v, row0, row1, ctx->gf4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pot, nullptr, nullptr, nullptr, st));
    ctx->gf4_rows[0] = row0; ctx->gf4_rows[1] = row1; ctx->gf4_n = u->n;
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_sweep0_finish(rebxamd_ctx* ctx, const rebxamd_view* u, double c, int64_t row0, int64_t row1,
                                            const double* anx, const double* any_, const double* anz,
                                            const double* aox, const double* aoy, const double* aoz,
                                            const double* acx, const double* acy, const double* acz,
                                            double* outx, double* outy, double* outz, uint32_t* dev_flag, int mode, void* stream) {
    if (mode == REBXAMD_MODE_EXACT)
        return rebxamd_grfull_sweep(ctx, u, c, row0, row1, anx, any_, anz, aox, aoy, aoz, acx, acy, acz, outx, outy, outz, dev_flag, mode, stream);
    int rc = grfull_check(ctx, u, row0, row1);
    if (rc != REBXAMD_OK) return rc;
    if (!aox || !aoy || !aoz || !acx || !acy || !acz || !outx || !outy || !outz || !dev_flag) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL buffer");
    if (!ctx->gf4 || ctx->gf4_rows[0] != row0 || ctx->gf4_rows[1] != row1 || ctx->gf4_n != u->n)
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_grfull_sweep0_finish: no pair sums of rebxamd_grfull_pot_sweep0 for these rows are waiting");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_grfull_sweep0_pot_combine(true, make_view(u), row0, row1, ctx->gf4, acx, acy, acz, aox, aoy, aoz, outx, outy, outz, dev_flag, (hipStream_t)stream));
    ctx->gf4_n = -1;
    return REBXAMD_OK;
}

// The exchange of a row-sharded sweep in two small kernels around the driver's all-gather (see grfull_kernels.hip).
extern "C" int rebxamd_grfull_pack(rebxamd_ctx* ctx, const double* outx, const double* outy, const double* outz, int64_t row0, int64_t per, int64_t n,
                                   uint32_t* dev_flag, double* send, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!outx || !outy || !outz || !dev_flag || !send || row0 < 0 || per < 0 || n < 0) return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_grfull_pack: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_grfull_pack(outx, outy, outz, row0, per, n, dev_flag, send, (hipStream_t)stream));
    return REBXAMD_OK;
}
extern "C" int rebxamd_grfull_unpack(rebxamd_ctx* ctx, const double* recv, int world, int64_t per, double* outx, double* outy, double* outz,
                                     uint32_t* dev_flag, uint32_t* stop_dev, int32_t* vote_dev, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!stop_dev || !vote_dev || (recv ? (world < 1 || per < 1 || !outx || !outy || !outz) : !dev_flag))
        return fail(ctx, REBXAMD_ERR_ARG, "rebxamd_grfull_unpack: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_grfull_unpack(recv, world, per, outx, outy, outz, dev_flag, stop_dev, vote_dev, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_stop_word(rebxamd_ctx* ctx, const uint32_t* stop_dev) {
    if (!ctx) return REBXAMD_ERR_ARG;
    ctx->gf_stop = stop_dev;
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_newton(rebxamd_ctx* ctx, const rebxamd_view* u, int gravity_ignore_10,
                                     double* anx, double* any_, double* anz, double* a0x, double* a0y, double* a0z, void* stream) {
    int rc = grfull_check(ctx, u, 0, u ? u->n : 0);
    if (rc != REBXAMD_OK) return rc;
    if (!anx || !any_ || !anz || !a0x || !a0y || !a0z || (u->n > 0 && (!u->ax || !u->ay || !u->az)))
        return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_grfull_prepare(make_view(u), u->G, gravity_ignore_10, anx, any_, anz, a0x, a0y, a0z, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull_finish(rebxamd_ctx* ctx, const rebxamd_view* u, const double* ax, const double* ay, const double* az, void* stream) {
    if (!ctx) return REBXAMD_ERR_ARG;
    if (!u || u->n < 0 || !ax || !ay || !az || (u->n > 0 && (!u->ax || !u->ay || !u->az))) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_grfull_finish(make_view(u), ax, ay, az, (hipStream_t)stream));
    return REBXAMD_OK;
}

extern "C" int rebxamd_grfull(rebxamd_ctx* ctx, const rebxamd_view* u, double c, int gravity_ignore_10, int mode, int* sweeps, void* stream) {
    int rc = grfull_check(ctx, u, 0, u ? u->n : 0);
    if (rc != REBXAMD_OK) return rc;
    if (sweeps) *sweeps = 0;
    const long long n = u->n;
    if (n <= 0) return REBXAMD_OK;
    if (!u->ax || !u->ay || !u->az) return fail(ctx, REBXAMD_ERR_ARG, "gr_full: acceleration columns missing");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    // workspace: a_newton(3) pot(1) a_const(3) a_old(3) a_new(3) = 13 columns; default arithmetic: behind them the partial sums
    // of the pass that the first sweep and the potential sums share (they outlive the a_const pass, which uses the context's own)
    const bool fused = mode != REBXAMD_MODE_EXACT;
    rc = ensure(ctx, &ctx->gf, &ctx->gf_cap, (size_t)13 * (size_t)n * sizeof(double) + (fused ? grfull_partial_bytes(n, n) : 0), st);
    if (rc != REBXAMD_OK) return rc;
    double* w = ctx->gf;
    double *anx = w, *any_ = w + n, *anz = w + 2 * n, *pot = w + 3 * n;
    double *acx = w + 4 * n, *acy = w + 5 * n, *acz = w + 6 * n;
    double *aox = w + 7 * n, *aoy = w + 8 * n, *aoz = w + 9 * n;
    double *nwx = w + 10 * n, *nwy = w + 11 * n, *nwz = w + 12 * n;
    const View v = make_view(u);
    if (n <= grfull_small_limit() && ctx->grfull_one_launch) {
        // small systems: every pass and every sweep in one launch of one workgroup (same functions, same bits)
        HIP_TRY(launch_grfull_small(v, u->G, c, gravity_ignore_10, w, ctx->gf_flag, ctx->status, !fused, st));
        if (sweeps) {
            unsigned h[2] = {0u, 0u};
            { const int rb = read_back(ctx, h, ctx->gf_flag, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
            *sweeps = (int)h[1];
        }
        return REBXAMD_OK;
    }
    HIP_TRY(launch_grfull_prepare(v, u->G, gravity_ignore_10, anx, any_, anz, aox, aoy, aoz, st));
    double* const part4 = w + 13 * n;
    if (fused) {
        HIP_TRY(launch_grfull_sweep0_pot(v, u->G, c, 0, n, anx, any_, anz, part4, st));
        HIP_TRY(launch_grfull_sweep0_pot_combine(false, v, 0, n, part4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pot, nullptr, nullptr, nullptr, st));
    } else {
        rc = rebxamd_grfull_pot(ctx, u, c, 0, n, pot, mode, stream);
        if (rc != REBXAMD_OK) return rc;
    }
    rc = rebxamd_grfull_const(ctx, u, c, 0, n, pot, acx, acy, acz, mode, stream);
    if (rc != REBXAMD_OK) return rc;
    // All ten sweeps (reference gr_full.c:173: the bound is the literal 10) are queued at once: the stopping rule is
    // evaluated on the device, and the kernels of a sweep queued past the one that met it return at once.  Sweep k reads
    // the iterate in buffer k & 1 and writes the other one.  The host waits only if the caller asks for the sweep count.
    unsigned* const ctl = ctx->gf_flag;
    HIP_TRY(hipMemsetAsync(ctl, 0, 4 * sizeof(unsigned), st));
    double* const b0[3] = {aox, aoy, aoz};
    double* const b1[3] = {nwx, nwy, nwz};
    const unsigned* const caller_stop = ctx->gf_stop;
    ctx->gf_stop = ctl;
    for (int k = 0; k < 10 && rc == REBXAMD_OK; k++) {
        double* const* src = (k & 1) ? b1 : b0;
        double* const* dst = (k & 1) ? b0 : b1;
        if (k == 0 && fused) {
            // the first sweep's pair sums are waiting in part4: add a_const, vote
            if (launch_grfull_sweep0_pot_combine(true, v, 0, n, part4, acx, acy, acz, src[0], src[1], src[2], dst[0], dst[1], dst[2], ctl + 2, st) != hipSuccess)
                rc = fail(ctx, REBXAMD_ERR_HIP, "gr_full: first sweep");
        } else rc = rebxamd_grfull_sweep(ctx, u, c, 0, n, anx, any_, anz, src[0], src[1], src[2], acx, acy, acz, dst[0], dst[1], dst[2], ctl + 2, mode, stream);
        if (rc == REBXAMD_OK && launch_grfull_vote(ctl, st) != hipSuccess) rc = fail(ctx, REBXAMD_ERR_HIP, "gr_full: vote kernel");
        // Small systems, where a sweep takes less time than the launches of the sweeps queued behind it cost when they find the
        // stop word set (three kernels each, a few microseconds apiece): the host looks at the stop word after every fourth sweep
        // instead of queueing all ten at once.
        if (rc == REBXAMD_OK && n <= 16384 && (k + 1) % 4 == 0 && k + 1 < 10) {
            unsigned h[2] = {0u, 0u};
            rc = read_back(ctx, h, ctl, sizeof(h), st);
            if (rc == REBXAMD_OK && h[0] != 0u) break;
        }
    }
    ctx->gf_stop = caller_stop;
    if (rc != REBXAMD_OK) return rc;
    HIP_TRY(launch_grfull_finish_select(v, b0, b1, ctl, ctx->status, st));
    if (sweeps) {
        unsigned h[2] = {0u, 0u};
        { const int rb = read_back(ctx, h, ctl, sizeof(h), st); if (rb != REBXAMD_OK) return rb; }
        *sweeps = (int)h[1];
    }
    return REBXAMD_OK;
}

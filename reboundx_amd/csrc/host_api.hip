// host_api.hip -- the plugin API (group A of reboundx_amd.h): the reference's object model
// (linked lists of named, typed parameters hanging off particles and forces), its force
// factory, and the additional-force dispatch -- with the arithmetic handed to the device.
//
// The linked lists stay the single source of truth, exactly as laid out by the reference, so
// C programs and ctypes bindings can keep walking them.  What changes is who reads them on
// the hot path: instead of 1-5 strcmp walks per particle per force per call, the lists are
// walked once per change of their structure and flattened into fp64 columns (one marker value =
// not set) plus short lists of source particles, which live on the GPU.  The walk also leaves a
// flat table of pointers to the values, through which every call reads them again (they may be
// overwritten in place); a call then moves positions, velocities, masses and accelerations, and
// the parameter columns in which a value changed.
#include "internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cfloat>
#include <climits>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <unistd.h>

#define STRINGIFY2(s) #s
#define STRINGIFY(s) STRINGIFY2(s)
#ifndef REBXGITHASH
#define REBXGITHASH rebxamd0000000000000000000000000000000001
#endif

extern "C" {
const char* rebx_build_str = __DATE__ " " __TIME__;
const char* rebx_version_str = "3.0.4";             // the reference version whose API this library speaks
const char* rebx_githash_str = STRINGIFY(REBXGITHASH);
}

// ------------------------------------------------------------------ standalone message channel
#ifndef REBXAMD_WITH_REBOUND
static void push_message(struct reb_simulation* r, char kind, const char* msg) {
    if (!r) { fprintf(stderr, "%s\n", msg); return; }
    char** grown = (char**)realloc(r->messages, sizeof(char*) * (size_t)(r->n_messages + 1));
    if (!grown) return;
    r->messages = grown;
    const size_t len = strlen(msg);
    char* copy = (char*)malloc(len + 2);
    if (!copy) return;
    copy[0] = kind;
    memcpy(copy + 1, msg, len + 1);
    r->messages[r->n_messages++] = copy;
}
extern "C" void reb_error(struct reb_simulation* r, const char* msg) { push_message(r, 'e', msg); }
extern "C" void reb_warning(struct reb_simulation* r, const char* msg) { push_message(r, 'w', msg); }
extern "C" int rebxamd_pop_message(struct reb_simulation* r, char* buf, int len) {
    if (!r || r->n_messages <= 0 || !buf || len <= 0) return 0;
    char* m = r->messages[0];
    const int kind = m[0] == 'e' ? 2 : 1;
    strncpy(buf, m + 1, (size_t)len - 1);
    buf[len - 1] = '\0';
    free(m);
    memmove(r->messages, r->messages + 1, sizeof(char*) * (size_t)(r->n_messages - 1));
    r->n_messages--;
    return kind;
}
extern "C" void rebxamd_clear_messages(struct reb_simulation* r) {
    if (!r) return;
    for (int i = 0; i < r->n_messages; i++) free(r->messages[i]);
    free(r->messages);
    r->messages = nullptr;
    r->n_messages = 0;
}
#endif

using rebxamd::WARN_GR_ITER;
using rebxamd::WARN_GRFULL_SWEEPS;
using rebxamd::WARN_IM_ITER;

// ------------------------------------------------------------------ per-extras device state
namespace {

enum { H_X, H_Y, H_Z, H_VX, H_VY, H_VZ, H_M, H_AX, H_AY, H_AZ, H_NDYN };
enum { P_BETA, P_RT, P_K1, P_TA, P_TE, P_TI, P_TM, P_TOM, P_TON, P_NPAR };
static const char* const kParamName[P_NPAR] = {"beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc", "tau_mass", "tau_omega", "tau_Omega"};

// Where the reference would read a value on every call: the pointer a list walk ends at.  The walk is
// cached (the structure of the lists changes rarely), the values are read through these on every call.
struct ValueRef { int i, k; const double* p; };        // particle, dense column, value
struct SourceJ { int idx; const double *J, *Req; };
struct SourceC { int idx; const double *A, *gamma; };

// what a NaN-valued parameter is uploaded as: the one NaN the kernels do not read as "not set", so that it
// propagates through them as it does through the reference's arithmetic
constexpr unsigned long long kNaNValueBits = REBXAMD_NAN_VALUE_BITS;
inline unsigned long long bits_of(double v) { return __builtin_bit_cast(unsigned long long, v); }
inline double as_column_value(double v) { return v == v ? v : __builtin_bit_cast(double, kNaNValueBits); }

struct Bridge {
    rebxamd_ctx* ctx = nullptr;
    int mode = REBXAMD_MODE_FAST;
    bool in_place = true;                   // small systems are evaluated in the pinned host columns (rebxamd_set_in_place)
    unsigned long long generation = 1;      // bumped by every API call that can change a parameter list
    // staging
    size_t cap = 0;
    double* h_slab = nullptr;               // pinned host slab and device slab holding the columns below, `cap` apart
    double* d_slab = nullptr;
    double* h_dyn[H_NDYN] = {};             // pinned host columns
    double* d_dyn[H_NDYN] = {};
    double* hd_dyn[H_NDYN] = {};            // the pinned host columns as the device addresses them
    double* h_par[P_NPAR] = {};             // pageable host columns (rebuilt rarely)
    double* d_par[P_NPAR] = {};
    bool par_used[P_NPAR] = {};             // column has at least one value set
    std::vector<ValueRef> refs;             // every dense parameter found in a particle's list, by particle
    std::vector<double> bulk[P_NPAR];       // columns given whole (rebxamd_set_param_column); NaN = not set
    // validity of the flattened parameters
    unsigned long long built_generation = 0;
    int built_n = -1;
    unsigned long long built_checksum = 0;
    // sparse per-particle parameters (sources)
    std::vector<int> radiation_sources, tides_primaries;
    int first_primary = -1;
    std::vector<SourceJ> j2, j4;
    std::vector<SourceC> central;
    double timings[4] = {0, 0, 0, 0};
    std::vector<hipEvent_t> block_done;     // one per block of the download/unpack pipeline
    size_t mass_synced = 0;                 // leading entries of the mass column that the device already holds
};

std::mutex g_mu;
std::unordered_map<const rebx_extras*, Bridge*> g_bridges;

Bridge* bridge_of(const rebx_extras* rebx, bool create) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_bridges.find(rebx);
    if (it != g_bridges.end()) return it->second;
    if (!create) return nullptr;
    Bridge* b = new Bridge();
    g_bridges[rebx] = b;
    return b;
}

// A bridge that does not exist yet has nothing cached: it is created, with everything stale, by
// the first call that needs the device.  (Creating one here would also leak it when a particle's
// list is freed after its extras object.)
void touch(const rebx_extras* rebx) {
    if (Bridge* b = bridge_of(rebx, false)) b->generation++;
}

void release_buffers(Bridge* b) {
    for (hipEvent_t e : b->block_done) (void)hipEventDestroy(e);
    b->block_done.clear();
    if (b->h_slab) (void)hipHostFree(b->h_slab);
    if (b->d_slab) (void)hipFree(b->d_slab);
    b->h_slab = b->d_slab = nullptr;
    for (int k = 0; k < H_NDYN; k++) b->h_dyn[k] = b->d_dyn[k] = b->hd_dyn[k] = nullptr;
    for (int k = 0; k < P_NPAR; k++) {
        free(b->h_par[k]);
        if (b->d_par[k]) (void)hipFree(b->d_par[k]);
        b->h_par[k] = b->d_par[k] = nullptr;
    }
    b->cap = 0;
    b->built_n = -1;
    b->mass_synced = 0;
}

void drop_bridge(const rebx_extras* rebx) {
    Bridge* b = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_bridges.find(rebx);
        if (it == g_bridges.end()) return;
        b = it->second;
        g_bridges.erase(it);
    }
    if (b->ctx) {
        release_buffers(b);
        rebxamd_destroy(b->ctx);
    }
    delete b;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Host threads for the array-of-structs <-> column conversions (REBXAMD_HOST_THREADS, default 16,
// at most 64; never more than the machine has): one per 16384 particles of the piece at hand, so that a helper's
// wake-up (a few microseconds: they are parked, not started, see HostPool) is small against its share of the work.
int host_threads(int N) {
    static const int configured = [] {
        const char* e = getenv("REBXAMD_HOST_THREADS");
        int t = e ? atoi(e) : 16;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && t > hw) t = hw;
        return t < 1 ? 1 : (t > 64 ? 64 : t);
    }();
    const int wanted = N / 16384;
    return wanted < 2 ? 1 : (wanted < configured ? wanted : configured);
}

// Particles per block of the pack/upload and download/unpack pipelines: about an eighth of the array, so that from a
// hundred thousand particles on the copy engine works underneath the host threads, between 64 Ki (below it a block's
// copies are all latency) and 1 Mi particles (above it nothing more is hidden); a multiple of 8192, which keeps every
// block of a column 16-byte aligned.
int pipeline_block(int N) {
    const long long eighth = ((long long)N / 8 + 8191) / 8192 * 8192;
    return (int)std::min<long long>(1 << 20, std::max<long long>(1 << 16, eighth));
}

// Systems this small move as one slab per direction and skip the synchronisations that only
// separate the phases of rebxamd_last_timings (their enqueue times are reported instead: what
// matters at this size is the latency of the call).
// (Buffers that once held a large system keep their size: then the columns are copied one by one.)
int small_limit() {
    static const int limit = [] { const char* e = getenv("REBXAMD_SMALL_LIMIT"); const int v = e ? atoi(e) : 32768; return v < 1 ? 1 : v; }();
    return limit;
}
bool small_system(const Bridge* b, int N) { return N <= small_limit() && b->cap <= 2 * (size_t)small_limit(); }

// Helper threads of the conversions, started once and parked on a condition variable between calls (starting a thread
// costs tens of microseconds, a conversion of 10^5 particles a hundred).  One job at a time: a caller that finds the
// pool busy (several simulations stepping on several host threads) starts threads of its own.  The pool is never
// destroyed -- its threads must not outlive what they wait on at process exit -- and a forked child, which inherits
// the object but none of the threads, starts afresh.
class HostPool {
  public:
    static HostPool& get() {
        static HostPool* pool = nullptr;
        static std::mutex guard;
        std::lock_guard<std::mutex> lock(guard);
        if (!pool || pool->pid_ != getpid()) pool = new HostPool;
        return *pool;
    }
    // call(arg, k) for k = 1 .. nt-1 on helpers, k = 0 on the caller; false (nothing done) when the pool is busy
    bool run(int nt, void (*call)(void*, int), void* arg) {
        std::unique_lock<std::mutex> busy(run_mu_, std::try_to_lock);
        if (!busy.owns_lock()) return false;
        {
            std::unique_lock<std::mutex> lk(mu_);
            while ((int)workers_.size() < nt - 1) {
                const int k = (int)workers_.size() + 1;
                workers_.emplace_back([this, k, seen = gen_] { work(k, seen); });
                workers_.back().detach();
            }
            call_ = call; arg_ = arg; njobs_ = nt; pending_ = nt - 1; gen_++;
        }
        cv_work_.notify_all();
        call(arg, 0);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this] { return pending_ == 0; });
        return true;
    }

  private:
    HostPool() : pid_(getpid()) {}
    void work(int k, unsigned long long seen) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (k >= njobs_) continue;
            void (*const call)(void*, int) = call_;
            void* const arg = arg_;
            lk.unlock();
            call(arg, k);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
    const pid_t pid_;
    std::mutex run_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    void (*call_)(void*, int) = nullptr;
    void* arg_ = nullptr;
    int njobs_ = 0, pending_ = 0;
    unsigned long long gen_ = 0;
};

// fn(lo, hi, k) on nt contiguous chunks of [0, N), chunk 0 on the calling thread (every chunk is called, empty ones too)
template <class Fn> void parallel_chunks(int N, int nt, Fn fn) {
    if (nt <= 1) { fn(0, N, 0); return; }
    const long long per = ((long long)N + nt - 1) / nt;
    auto chunk = [&](int k) {
        const int lo = (int)std::min<long long>(N, k * per), hi = (int)std::min<long long>(N, (k + 1) * per);
        fn(lo, hi, k);
    };
    using Chunk = decltype(chunk);
    if (HostPool::get().run(nt, [](void* c, int k) { (*static_cast<Chunk*>(c))(k); }, &chunk)) return;
    std::vector<std::thread> own;
    own.reserve((size_t)nt - 1);
    for (int k = 1; k < nt; k++) own.emplace_back([&chunk, k] { chunk(k); });
    chunk(0);
    for (std::thread& t : own) t.join();
}

}  // namespace

// ------------------------------------------------------------------ lists (reference src/linkedlist.c)
static void list_push_front(struct rebx_node** head, struct rebx_node* node) {      // :33-37
    node->next = *head;
    *head = node;
}

static int list_remove_object(struct rebx_node** head, void* object) {              // :71-95
    for (struct rebx_node** link = head; *link; link = &(*link)->next) {
        if ((*link)->object == object) {
            struct rebx_node* dead = *link;
            *link = dead->next;
            free(dead);
            return 1;
        }
    }
    return 0;
}

extern "C" int rebx_len(struct rebx_node* head) {                                   // :97-106
    int n = 0;
    for (; head; head = head->next) n++;
    return n;
}

// ------------------------------------------------------------------ errors, memory
extern "C" void rebx_error(struct rebx_extras* rebx, const char* const msg) {       // src/core.c:1003-1010
    if (!rebx || rebx->sim == NULL) {
        fprintf(stderr, "REBOUNDx Error: A Simulation is no longer attached to this REBOUNDx extras instance. Most likely the Simulation has been freed.\n");
        return;
    }
    reb_error(rebx->sim, msg);
}

static void* checked_malloc(struct rebx_extras* rebx, size_t bytes) {               // src/core.c:730-738
    void* p = malloc(bytes);
    if (!p && bytes > 0) rebx_error(rebx, "REBOUNDx Error: Could not allocate memory.\n");
    return p;
}

static char* dup_name(struct rebx_extras* rebx, const char* name) {
    char* c = (char*)checked_malloc(rebx, strlen(name) + 1);
    if (c) strcpy(c, name);
    return c;
}

static struct rebx_node* new_node(struct rebx_extras* rebx, void* object) {         // src/core.c:918-926
    struct rebx_node* n = (struct rebx_node*)checked_malloc(rebx, sizeof(*n));
    if (n) { n->object = object; n->next = NULL; }
    return n;
}

static void free_param(struct rebx_param* p, bool owns_value) {                     // src/core.c:740-751, :792-797
    free(p->name);
    // numeric values were allocated by the setters; pointers and forces are borrowed
    if (owns_value && (p->type == REBX_TYPE_INT || p->type == REBX_TYPE_DOUBLE || p->type == REBX_TYPE_UINT32)) free(p->value);
    free(p);
}

static void free_param_list(struct rebx_node** ap) {                                // src/core.c:753-762
    struct rebx_node* cur = *ap;
    while (cur) {
        struct rebx_node* next = cur->next;
        free_param((struct rebx_param*)cur->object, true);
        free(cur);
        cur = next;
    }
    *ap = NULL;
}

extern "C" void rebx_free_particle_ap(struct reb_particle* p) {                     // src/core.c:764-766
    struct rebx_node* head = (struct rebx_node*)p->ap;
    free_param_list(&head);
    p->ap = NULL;
    if (p->sim && p->sim->extras) touch((const rebx_extras*)p->sim->extras);
}

// ------------------------------------------------------------------ parameter registry and access
extern "C" struct rebx_param* rebx_get_param_struct(struct rebx_extras*, struct rebx_node* ap, const char* const param_name) {
    for (; ap; ap = ap->next) {                                                     // src/core.c:601-612
        struct rebx_param* p = (struct rebx_param*)ap->object;
        if (strcmp(p->name, param_name) == 0) return p;
    }
    return NULL;
}

extern "C" void* rebx_get_param(struct rebx_extras* rebx, struct rebx_node* ap, const char* const param_name) {
    struct rebx_param* p = rebx_get_param_struct(rebx, ap, param_name);             // src/core.c:614-622
    return p ? p->value : NULL;
}

extern "C" enum rebx_param_type rebx_get_type(struct rebx_extras* rebx, const char* name) {
    struct rebx_param* p = rebx_get_param_struct(rebx, rebx->registered_params, name);   // src/core.c:958-966
    return p ? p->type : REBX_TYPE_NONE;
}

static struct rebx_param* new_param(struct rebx_extras* rebx, const char* name, enum rebx_param_type type) {
    struct rebx_param* p = (struct rebx_param*)checked_malloc(rebx, sizeof(*p));    // src/core.c:928-945
    if (!p) return NULL;
    p->type = type;
    p->value = NULL;
    p->name = dup_name(rebx, name);
    if (!p->name) { free(p); return NULL; }
    return p;
}

extern "C" void rebx_register_param(struct rebx_extras* const rebx, const char* name, enum rebx_param_type type) {
    if (rebx_get_type(rebx, name) != REBX_TYPE_NONE) {                              // src/core.c:94-117
        char str[300];
        snprintf(str, sizeof(str), "REBOUNDx Error: Parameter name '%s' already in registered list. Cannot add duplicates.\n", name);
        rebx_error(rebx, str);
        return;
    }
    struct rebx_param* p = new_param(rebx, name, type);
    if (!p) return;
    struct rebx_node* n = new_node(rebx, p);
    if (!n) { free_param(p, false); return; }
    list_push_front(&rebx->registered_params, n);
}

extern "C" void rebx_register_default_params(struct rebx_extras* rebx) {            // src/core.c:51-92
    static const struct { const char* name; enum rebx_param_type type; } defaults[] = {
        {"c", REBX_TYPE_DOUBLE}, {"gr_source", REBX_TYPE_INT}, {"tau_mass", REBX_TYPE_DOUBLE}, {"force", REBX_TYPE_FORCE},
        {"particle", REBX_TYPE_POINTER}, {"Acentral", REBX_TYPE_DOUBLE}, {"gammacentral", REBX_TYPE_DOUBLE},
        {"max_iterations", REBX_TYPE_INT}, {"J2", REBX_TYPE_DOUBLE}, {"J4", REBX_TYPE_DOUBLE}, {"R_eq", REBX_TYPE_DOUBLE},
        {"coordinates", REBX_TYPE_INT}, {"p", REBX_TYPE_DOUBLE}, {"tau_a", REBX_TYPE_DOUBLE}, {"tau_e", REBX_TYPE_DOUBLE},
        {"tau_inc", REBX_TYPE_DOUBLE}, {"tau_omega", REBX_TYPE_DOUBLE}, {"tau_Omega", REBX_TYPE_DOUBLE}, {"primary", REBX_TYPE_INT},
        {"radiation_source", REBX_TYPE_INT}, {"beta", REBX_TYPE_DOUBLE}, {"tides_primary", REBX_TYPE_INT},
        {"R_tides", REBX_TYPE_DOUBLE}, {"k1", REBX_TYPE_DOUBLE}, {"integrator", REBX_TYPE_INT}, {"free_arrays", REBX_TYPE_POINTER},
        {"im_ps_final", REBX_TYPE_POINTER}, {"im_ps_prev", REBX_TYPE_POINTER}, {"im_ps_avg", REBX_TYPE_POINTER},
        {"rk2_k2", REBX_TYPE_POINTER}, {"rk4_k2", REBX_TYPE_POINTER}, {"rk4_k3", REBX_TYPE_POINTER},
        {"min_distance", REBX_TYPE_DOUBLE}, {"min_distance_from", REBX_TYPE_UINT32}, {"min_distance_orbit", REBX_TYPE_ORBIT},
        {"N_ephem", REBX_TYPE_INT}, {"N_ast", REBX_TYPE_INT}, {"geocentric", REBX_TYPE_INT}, {"outstate", REBX_TYPE_POINTER},
        {"n_out", REBX_TYPE_INT},
    };
    for (const auto& d : defaults) rebx_register_param(rebx, d.name, d.type);
}

// existing parameter of that name, or a fresh one pushed on the list (src/core.c:512-541)
static struct rebx_param* find_or_add(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const name) {
    if (apptr == NULL) {
        rebx_error(rebx, "REBOUNDx Error: Passed NULL apptr to rebx_add_param. See examples.\n");
        return NULL;
    }
    const enum rebx_param_type type = rebx_get_type(rebx, name);
    if (type == REBX_TYPE_NONE) {
        char str[300];
        snprintf(str, sizeof(str), "REBOUNDx Error: Need to register parameter name '%s' before using it. See examples.\n", name);
        rebx_error(rebx, str);
        return NULL;
    }
    struct rebx_param* p = rebx_get_param_struct(rebx, *apptr, name);
    if (p) return p;
    p = new_param(rebx, name, type);
    if (!p) return NULL;
    struct rebx_node* n = new_node(rebx, p);
    if (!n) { free_param(p, false); return NULL; }
    list_push_front(apptr, n);
    return p;
}

template <typename T>
static void set_numeric(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const name, T val) {
    struct rebx_param* p = find_or_add(rebx, apptr, name);                          // src/core.c:552-595
    if (!p) return;
    if (p->value == NULL) p->value = checked_malloc(rebx, sizeof(T));
    if (p->value) *(T*)p->value = val;
    touch(rebx);
}

extern "C" void rebx_set_param_double(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const name, double val) { set_numeric<double>(rebx, apptr, name, val); }
extern "C" void rebx_set_param_int(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const name, int val) { set_numeric<int>(rebx, apptr, name, val); }
extern "C" void rebx_set_param_uint32(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const name, uint32_t val) { set_numeric<uint32_t>(rebx, apptr, name, val); }
extern "C" void rebx_set_param_pointer(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const name, void* val) {
    struct rebx_param* p = find_or_add(rebx, apptr, name);                          // src/core.c:543-550
    if (!p) return;
    p->value = val;
    touch(rebx);
}

// ------------------------------------------------------------------ attach / detach / free
extern "C" void rebx_extras_cleanup(struct reb_simulation* sim) {                   // src/core.c:130-133
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    if (rebx) rebx->sim = NULL;
}

extern "C" void rebx_initialize(struct reb_simulation* sim, struct rebx_extras* rebx) {   // src/core.c:158-178
    rebx->sim = sim;
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return; }
    sim->extras = rebx;
    rebx->additional_forces = NULL;
    rebx->pre_timestep_modifications = NULL;
    rebx->post_timestep_modifications = NULL;
    rebx->allocated_forces = NULL;
    rebx->allocated_operators = NULL;
    rebx->registered_params = NULL;
    sim->free_particle_ap = rebx_free_particle_ap;
    sim->extras_cleanup = rebx_extras_cleanup;
    if (sim->additional_forces || sim->pre_timestep_modifications || sim->post_timestep_modifications) {
        reb_warning(sim, "REBOUNDx overwrites sim->additional_forces, sim->pre_timestep_modifications and sim->post_timestep_modifications whenever forces or operators that use them get added.  If you want to use REBOUNDx together with your own custom functions that use these callbacks, you should add them through REBOUNDx.");
    }
}

extern "C" struct rebx_extras* rebx_attach(struct reb_simulation* sim) {            // src/core.c:119-128
    if (sim == NULL) {
        fprintf(stderr, "REBOUNDx Error: Simulation pointer passed to rebx_attach was NULL.\n");
        return NULL;
    }
    struct rebx_extras* rebx = (struct rebx_extras*)malloc(sizeof(*rebx));
    if (!rebx) return NULL;
    rebx_initialize(sim, rebx);
    rebx_register_default_params(rebx);
    return rebx;
}

extern "C" void rebx_detach(struct reb_simulation* sim, struct rebx_extras* rebx) { // src/core.c:135-155
    if (sim == NULL) return;
    rebx->sim = NULL;
    if (sim->extras == rebx) {          // only un-hook what is still ours
        if (sim->additional_forces == rebx_additional_forces) sim->additional_forces = NULL;
        if (sim->pre_timestep_modifications == rebx_pre_timestep_modifications) sim->pre_timestep_modifications = NULL;
        if (sim->post_timestep_modifications == rebx_post_timestep_modifications) sim->post_timestep_modifications = NULL;
        if (sim->free_particle_ap == rebx_free_particle_ap) sim->free_particle_ap = NULL;
    }
}

static void free_force(struct rebx_extras* rebx, struct rebx_force* force) {        // src/core.c:768-778
    typedef void (*destructor)(struct rebx_extras*, struct rebx_force*);
    destructor free_arrays = (destructor)rebx_get_param(rebx, force->ap, "free_arrays");
    if (free_arrays) free_arrays(rebx, force);
    free(force->name);
    free_param_list(&force->ap);
    free(force);
}

extern "C" void rebx_free_pointers(struct rebx_extras* rebx) {                      // src/core.c:799-854
    if (rebx == NULL) return;
    rebx_detach(rebx->sim, rebx);
    for (struct rebx_node* cur = rebx->allocated_forces; cur;) {
        struct rebx_node* next = cur->next;
        free_force(rebx, (struct rebx_force*)cur->object);
        free(cur);
        cur = next;
    }
    rebx->allocated_forces = NULL;
    for (struct rebx_node* cur = rebx->additional_forces; cur;) {
        struct rebx_node* next = cur->next;
        free(cur);
        cur = next;
    }
    rebx->additional_forces = NULL;
    for (struct rebx_node** head : {&rebx->pre_timestep_modifications, &rebx->post_timestep_modifications}) {
        for (struct rebx_node* cur = *head; cur;) {
            struct rebx_node* next = cur->next;
            free(cur->object);          // the step; its operator is owned by allocated_operators
            free(cur);
            cur = next;
        }
        *head = NULL;
    }
    for (struct rebx_node* cur = rebx->allocated_operators; cur;) {
        struct rebx_node* next = cur->next;
        struct rebx_operator* op = (struct rebx_operator*)cur->object;
        free(op->name);
        free_param_list(&op->ap);
        free(op);
        free(cur);
        cur = next;
    }
    rebx->allocated_operators = NULL;
    for (struct rebx_node* cur = rebx->registered_params; cur;) {
        struct rebx_node* next = cur->next;
        free_param((struct rebx_param*)cur->object, false);
        free(cur);
        cur = next;
    }
    rebx->registered_params = NULL;
    drop_bridge(rebx);                  // device buffers and context
}

extern "C" void rebx_free(struct rebx_extras* rebx) {                               // src/core.c:180-183
    rebx_free_pointers(rebx);
    free(rebx);
}

// ------------------------------------------------------------------ forces
extern "C" struct rebx_force* rebx_create_force(struct rebx_extras* const rebx, const char* name) {   // src/core.c:189-225
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return NULL; }
    struct rebx_force* force = (struct rebx_force*)checked_malloc(rebx, sizeof(*force));
    if (!force) return NULL;
    force->ap = NULL;
    force->sim = rebx->sim;
    force->force_type = REBX_FORCE_NONE;
    force->update_accelerations = NULL;
    force->name = NULL;
    if (name != NULL) {
        force->name = dup_name(rebx, name);
        if (!force->name) { free(force); return NULL; }
    }
    struct rebx_node* n = new_node(rebx, force);
    if (!n) { free(force->name); free(force); return NULL; }
    list_push_front(&rebx->allocated_forces, n);
    return force;
}

namespace {
struct Builtin {
    const char* name;
    void (*fn)(struct reb_simulation* const, struct rebx_force* const, struct reb_particle* const, const int);
    enum rebx_force_type type;
};
const Builtin kBuiltins[] = {                                                       // src/core.c:232-263
    {"gr", rebx_gr, REBX_FORCE_VEL},
    {"central_force", rebx_central_force, REBX_FORCE_POS},
    {"modify_orbits_forces", rebx_modify_orbits_forces, REBX_FORCE_VEL},
    {"gr_full", rebx_gr_full, REBX_FORCE_VEL},
    {"gravitational_harmonics", rebx_gravitational_harmonics, REBX_FORCE_POS},
    {"gr_potential", rebx_gr_potential, REBX_FORCE_POS},
    {"radiation_forces", rebx_radiation_forces, REBX_FORCE_VEL},
    {"tides_precession", rebx_tides_precession, REBX_FORCE_POS},
};
const Builtin* builtin_of(const struct rebx_force* f) {
    for (const Builtin& b : kBuiltins)
        if (f->update_accelerations == b.fn) return &b;
    return nullptr;
}
}  // namespace

extern "C" int rebx_remove_force(struct rebx_extras* rebx, struct rebx_force* force) {      // src/core.c:654-662
    const int allocated = list_remove_object(&rebx->allocated_forces, force);
    const int success = list_remove_object(&rebx->additional_forces, force);
    if (allocated) free_force(rebx, force);
    touch(rebx);
    return success;
}

extern "C" struct rebx_force* rebx_load_force(struct rebx_extras* const rebx, const char* name) {     // src/core.c:227-277
    struct rebx_force* force = rebx_create_force(rebx, name);
    if (!force) return NULL;
    for (const Builtin& b : kBuiltins) {
        if (strcmp(name, b.name) == 0) {
            force->update_accelerations = b.fn;
            force->force_type = b.type;
            return force;
        }
    }
    char str[300];
    snprintf(str, sizeof(str), "REBOUNDx error: Force '%s' not found in REBOUNDx library.\n", name);
    rebx_error(rebx, str);
    rebx_remove_force(rebx, force);
    return NULL;
}

extern "C" int rebx_add_force(struct rebx_extras* rebx, struct rebx_force* force) {         // src/core.c:371-409
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return 0; }
    if (force == NULL) { rebx_error(rebx, "REBOUNDx error: Passed NULL pointer to rebx_add_force.\n"); return 0; }
    if (force->update_accelerations == NULL) {
        rebx_error(rebx, "REBOUNDx error: Need to set update_accelerations function pointer on force before calling rebx_add_force. See custom effects example.\n");
        return 0;
    }
    if (force->force_type == REBX_FORCE_NONE) {
        rebx_error(rebx, "REBOUNDx error: Need to set force_type field on force before calling rebx_add_force. See custom effects example.\n");
        return 0;
    }
    if (force->force_type == REBX_FORCE_VEL) rebx->sim->force_is_velocity_dependent = 1;
    struct rebx_node* n = new_node(rebx, force);
    if (!n) return 0;
    list_push_front(&rebx->additional_forces, n);       // head insertion: forces run in reverse order of addition
    if (rebx->sim->additional_forces != NULL && rebx->sim->additional_forces != rebx_additional_forces) {
        reb_warning(rebx->sim, "REBOUNDx Warning: additional_forces was set and is being overwritten by REBOUNDx. To incorporate both, you can add your own custom effects through REBOUNDx.\n");
    }
    rebx->sim->additional_forces = rebx_additional_forces;
    touch(rebx);
    return 1;
}

extern "C" struct rebx_force* rebx_get_force(struct rebx_extras* const rebx, const char* const name) {    // src/core.c:624-635
    for (struct rebx_node* cur = rebx->allocated_forces; cur; cur = cur->next) {
        struct rebx_force* f = (struct rebx_force*)cur->object;
        if (f->name && strcmp(f->name, name) == 0) return f;
    }
    return NULL;
}

extern "C" void rebx_reset_accelerations(struct reb_particle* const ps, const int N) {      // src/core.c:860-866
    for (int i = 0; i < N; i++) { ps[i].ax = 0.; ps[i].ay = 0.; ps[i].az = 0.; }
}

extern "C" double rebx_rad_calc_beta(const double G, const double c, const double source_mass, const double source_luminosity,
                                     const double radius, const double density, const double Q_pr) {        // src/radiation_forces.c:119-121
    return 3. * source_luminosity * Q_pr / (16. * 3.14159265358979323846 * G * source_mass * c * density * radius);
}
extern "C" double rebx_rad_calc_particle_radius(const double G, const double c, const double source_mass, const double source_luminosity,
                                                const double beta, const double density, const double Q_pr) { // src/radiation_forces.c:122-124
    return 3. * source_luminosity * Q_pr / (16. * 3.14159265358979323846 * G * source_mass * c * density * beta);
}

// src/central_force.c:143-152: the normalisation A of a central force a = A r^gamma that turns the pericentre of
// p's orbit about `primary` at the rate pomegadot.  The reference takes the separation d and the mean motion n from
// REBOUND's reb_tools_particle_to_orbit (mu = G (m_p + m_primary), a = -mu / (v^2 - 2 mu / d), n = sign(a) sqrt(|mu / a^3|));
// an orbit REBOUND calls undefined (primary without mass, zero separation) comes back as NaN there, and so it does here.
extern "C" double rebx_central_force_Acentral(const struct reb_particle p, const struct reb_particle primary, const double pomegadot, const double gamma) {
    struct reb_simulation* const sim = p.sim;
    const double G = sim ? sim->G : 1.;
    const double TINY = 1.E-308;
    double d = NAN, n = NAN;
    if (primary.m > TINY) {
        const double mu = G * (p.m + primary.m);
        const double dx = p.x - primary.x, dy = p.y - primary.y, dz = p.z - primary.z;
        const double dvx = p.vx - primary.vx, dvy = p.vy - primary.vy, dvz = p.vz - primary.vz;
        const double sep = sqrt(dx * dx + dy * dy + dz * dz);
        if (sep > TINY) {
            const double vsquared = dvx * dvx + dvy * dvy + dvz * dvz;
            const double a = -mu / (vsquared - 2. * (mu / sep));
            d = sep;
            n = a / fabs(a) * sqrt(fabs(mu / (a * a * a)));
        }
    }
    if (fabs(gamma + 2.) < DBL_EPSILON) {       // the precession vanishes for a force law r^-2: A would diverge
        const char* const msg = "Precession vanishes for force law varying as r^-2, so can't initialize Acentral from a precession rate for gamma=-2)\n";
        if (sim) reb_error(sim, msg); else fputs(msg, stderr);
        return 0.;
    }
    return G * primary.m * pomegadot / (1. + gamma / 2.) / pow(d, gamma + 2.) / n;
}

// ------------------------------------------------------------------ extensions
extern "C" void rebxamd_invalidate_params(struct rebx_extras* rebx) { touch(rebx); }
static int dense_index(const char* name) {
    for (int k = 0; k < P_NPAR; k++)
        if (name && strcmp(name, kParamName[k]) == 0) return k;
    return -1;
}
extern "C" int rebxamd_set_param_column(struct rebx_extras* rebx, const char* name, const double* values, int N) {
    if (!rebx) return 0;
    const int k = dense_index(name);
    char str[300];
    if (k < 0) {
        snprintf(str, sizeof(str), "REBOUNDx (MI355X) Error: '%s' is not a per-particle parameter that can be given as a column.\n", name ? name : "(null)");
        rebx_error(rebx, str);
        return 0;
    }
    if (rebx_get_type(rebx, name) != REBX_TYPE_DOUBLE) {        // as the per-particle setter insists (src/core.c:512-541)
        snprintf(str, sizeof(str), "REBOUNDx Error: Need to register parameter name '%s' before using it. See examples.\n", name);
        rebx_error(rebx, str);
        return 0;
    }
    if (N < 0) { rebx_error(rebx, "REBOUNDx (MI355X) Error: negative length passed to rebxamd_set_param_column.\n"); return 0; }
    Bridge* b = bridge_of(rebx, true);
    if (values && N > 0) b->bulk[k].assign(values, values + N);
    else std::vector<double>().swap(b->bulk[k]);
    b->generation++;
    return 1;
}
extern "C" const double* rebxamd_get_param_column(struct rebx_extras* rebx, const char* name, int* N) {
    const int k = dense_index(name);
    Bridge* b = rebx ? bridge_of(rebx, false) : nullptr;
    const bool have = b && k >= 0 && !b->bulk[k].empty();
    if (N) *N = have ? (int)b->bulk[k].size() : 0;
    return have ? b->bulk[k].data() : nullptr;
}
extern "C" void rebxamd_set_mode(struct rebx_extras* rebx, int mode) {
    if (Bridge* b = bridge_of(rebx, true)) b->mode = mode ? REBXAMD_MODE_EXACT : REBXAMD_MODE_FAST;
}
extern "C" void rebxamd_set_in_place(struct rebx_extras* rebx, int on) {
    if (Bridge* b = bridge_of(rebx, true)) b->in_place = on != 0;
}
extern "C" void rebxamd_last_timings(struct rebx_extras* rebx, double out_ms[4]) {
    Bridge* b = bridge_of(rebx, false);
    for (int k = 0; k < 4; k++) out_ms[k] = b ? b->timings[k] : 0.;
}

// ------------------------------------------------------------------ host AoS <-> device columns
namespace {

bool ensure_capacity(struct rebx_extras* rebx, Bridge* b, size_t n) {
    if (!b->ctx) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (rebxamd_create(dev, &b->ctx) != REBXAMD_OK) {
            rebx_error(rebx, "REBOUNDx (MI355X) Error: no usable GPU: the additional-force path has no CPU fallback.\n");
            b->ctx = nullptr;
            return false;
        }
    }
    if (n <= b->cap) return true;
    release_buffers(b);
    const size_t cap = (n + n / 8 + 64 + 1) & ~(size_t)1;       // even: every column stays 16-byte aligned
    // one slab each side, so that small systems move in one copy per direction
    if (hipHostMalloc((void**)&b->h_slab, H_NDYN * cap * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&b->d_slab, H_NDYN * cap * sizeof(double)) != hipSuccess) {
        rebx_error(rebx, "REBOUNDx (MI355X) Error: could not allocate staging buffers.\n");
        release_buffers(b);
        return false;
    }
    double* mapped = nullptr;
    if (hipHostGetDevicePointer((void**)&mapped, b->h_slab, 0) != hipSuccess) mapped = nullptr;
    for (int k = 0; k < H_NDYN; k++) { b->h_dyn[k] = b->h_slab + k * cap; b->d_dyn[k] = b->d_slab + k * cap; b->hd_dyn[k] = mapped ? mapped + k * cap : nullptr; }
    for (int k = 0; k < P_NPAR; k++) {
        b->h_par[k] = (double*)malloc(cap * sizeof(double));
        if (!b->h_par[k] || hipMalloc((void**)&b->d_par[k], cap * sizeof(double)) != hipSuccess) {
            rebx_error(rebx, "REBOUNDx (MI355X) Error: could not allocate parameter columns.\n");
            release_buffers(b);
            return false;
        }
    }
    b->cap = cap;
    return true;
}

// Flattens every particle's parameter list: dense columns (rebxamd::unset_value() = not set) and the
// short lists of source-type parameters.  One walk per list, no matter how many forces are stacked.
// A column starts from the one given whole, if any; what a particle's own list holds wins over it.
void flatten_params(struct rebx_extras* rebx, Bridge* b, const struct reb_particle* ps, int N) {
    struct Local {
        std::vector<ValueRef> refs;
        std::vector<int> radiation_sources, tides_primaries;
        std::vector<SourceJ> j2, j4;
        std::vector<SourceC> central;
        int first_primary = -1;
    };
    const int nt = host_threads(N);
    std::vector<Local> local((size_t)nt);
    // each thread walks the lists of its own particles; the pieces are merged in index order afterwards
    parallel_chunks(N, nt, [&](int lo, int hi, int t) {
        Local& L = local[(size_t)t];
        for (int i = lo; i < hi; i++) {
            if (!ps[i].ap) continue;
            const double *J2 = nullptr, *J4 = nullptr, *Req = nullptr, *A = nullptr, *gam = nullptr;
            unsigned seen = 0;          // dense names already met in this list: the first match in list order wins, as a linear search finds it
            for (const struct rebx_node* node = (const struct rebx_node*)ps[i].ap; node; node = node->next) {
                const struct rebx_param* p = (const struct rebx_param*)node->object;
                const char* name = p->name;
                bool dense = false;
                for (int k = 0; k < P_NPAR; k++) {
                    if (strcmp(name, kParamName[k]) == 0) {
                        if (!(seen >> k & 1u)) {
                            seen |= 1u << k;
                            if (p->value) L.refs.push_back({i, k, (const double*)p->value});
                        }
                        dense = true;
                        break;
                    }
                }
                if (dense) continue;
                if (!strcmp(name, "J2")) { if (!J2) J2 = (const double*)p->value; }
                else if (!strcmp(name, "J4")) { if (!J4) J4 = (const double*)p->value; }
                else if (!strcmp(name, "R_eq")) { if (!Req) Req = (const double*)p->value; }
                else if (!strcmp(name, "Acentral")) { if (!A) A = (const double*)p->value; }
                else if (!strcmp(name, "gammacentral")) { if (!gam) gam = (const double*)p->value; }
                else if (!strcmp(name, "radiation_source")) { if (p->value && (L.radiation_sources.empty() || L.radiation_sources.back() != i)) L.radiation_sources.push_back(i); }
                else if (!strcmp(name, "tides_primary")) { if (p->value && (L.tides_primaries.empty() || L.tides_primaries.back() != i)) L.tides_primaries.push_back(i); }
                else if (!strcmp(name, "primary")) { if (p->value && L.first_primary < 0) L.first_primary = i; }
            }
            if (J2 && Req) L.j2.push_back({i, J2, Req});
            if (J4 && Req) L.j4.push_back({i, J4, Req});
            if (A && gam) L.central.push_back({i, A, gam});
        }
    });
    b->refs.clear();
    b->radiation_sources.clear(); b->tides_primaries.clear(); b->j2.clear(); b->j4.clear(); b->central.clear();
    b->first_primary = -1;
    size_t nrefs = 0;
    for (const Local& L : local) nrefs += L.refs.size();
    b->refs.reserve(nrefs);
    for (Local& L : local) {
        b->refs.insert(b->refs.end(), L.refs.begin(), L.refs.end());
        std::vector<ValueRef>().swap(L.refs);
        b->radiation_sources.insert(b->radiation_sources.end(), L.radiation_sources.begin(), L.radiation_sources.end());
        b->tides_primaries.insert(b->tides_primaries.end(), L.tides_primaries.begin(), L.tides_primaries.end());
        b->j2.insert(b->j2.end(), L.j2.begin(), L.j2.end());
        b->j4.insert(b->j4.end(), L.j4.begin(), L.j4.end());
        b->central.insert(b->central.end(), L.central.begin(), L.central.end());
        if (b->first_primary < 0) b->first_primary = L.first_primary;
    }
    // the columns somebody set: start from the column given whole (or "not set"), then the lists' values on top
    for (int k = 0; k < P_NPAR; k++) b->par_used[k] = !b->bulk[k].empty();
    for (const ValueRef& r : b->refs) b->par_used[r.k] = true;
    const double unset = rebxamd::unset_value();
    for (int k = 0; k < P_NPAR; k++) {
        if (!b->par_used[k]) continue;
        double* col = b->h_par[k];
        const double* given = b->bulk[k].data();
        const int ngiven = (int)std::min<size_t>(b->bulk[k].size(), (size_t)N);
        parallel_chunks(N, nt, [&](int lo, int hi, int) {
            for (int i = lo; i < hi; i++) col[i] = (i < ngiven && given[i] == given[i]) ? given[i] : unset;
        });
    }
    parallel_chunks((int)std::min<size_t>(b->refs.size(), (size_t)INT_MAX), nt, [&](int lo, int hi, int) {
        for (int r = lo; r < hi; r++) { const ValueRef& q = b->refs[(size_t)r]; b->h_par[q.k][q.i] = as_column_value(*q.p); }
    });
}

// Reads every cached value pointer again; marks the columns in which a value moved since the last call.
void refresh_values(Bridge* b, bool changed[P_NPAR]) {
    for (int k = 0; k < P_NPAR; k++) changed[k] = false;
    const int n = (int)b->refs.size();
    if (n == 0) return;
    const int nt = host_threads(n);
    bool moved[64][P_NPAR] = {};
    parallel_chunks(n, nt, [&](int lo, int hi, int t) {
        for (int r = lo; r < hi; r++) {
            const ValueRef& q = b->refs[(size_t)r];
            const double v = as_column_value(*q.p);
            double& slot = b->h_par[q.k][q.i];
            if (bits_of(slot) != bits_of(v)) { slot = v; moved[t][q.k] = true; }
        }
    });
    for (int t = 0; t < nt; t++)
        for (int k = 0; k < P_NPAR; k++) changed[k] = changed[k] || moved[t][k];
}

struct Pending { struct rebx_force* force; const Builtin* builtin; };

// Translates one built-in force into elementary ops; false (after reporting through the
// simulation's message channel) when a required parameter is missing.
bool ops_of_force(struct reb_simulation* sim, struct rebx_extras* rebx, Bridge* b, const Pending& pf, std::vector<rebxamd_op>& ops) {
    auto add = [&](int kind, int src, int iarg, double a, double bb) {
        rebxamd_op o; o.kind = kind; o.src = src; o.iarg = iarg; o.reserved = 0; o.a = a; o.b = bb; ops.push_back(o);
    };
    struct rebx_force* f = pf.force;
    const auto fn = pf.builtin->fn;
    const double* c = (const double*)rebx_get_param(rebx, f->ap, "c");
    if (fn == rebx_radiation_forces) {
        if (!c) { reb_error(sim, "Need to set speed of light in radiation_forces effect.  See examples in documentation.\n"); return false; }
        if (b->radiation_sources.empty()) add(REBXAMD_OP_RADIATION, 0, 0, *c, 0.);
        for (int s : b->radiation_sources) add(REBXAMD_OP_RADIATION, s, 0, *c, 0.);
    } else if (fn == rebx_gravitational_harmonics) {
        for (const SourceJ& s : b->j2) add(REBXAMD_OP_J2, s.idx, 0, *s.J, *s.Req);
        for (const SourceJ& s : b->j4) add(REBXAMD_OP_J4, s.idx, 0, *s.J, *s.Req);
    } else if (fn == rebx_gr) {
        if (!c) { reb_error(sim, "REBOUNDx Error: Need to set speed of light in gr effect.  See examples in documentation.\n"); return false; }
        const int* it = (const int*)rebx_get_param(rebx, f->ap, "max_iterations");
        add(REBXAMD_OP_GR, 0, it ? *it : 10, *c, 0.);
    } else if (fn == rebx_gr_potential) {
        if (!c) { reb_error(sim, "REBOUNDx Error: Need to set speed of light in gr effect.  See examples in documentation.\n"); return false; }
        add(REBXAMD_OP_GR_POTENTIAL, 0, 0, *c, 0.);
    } else if (fn == rebx_gr_full) {
        if (!c) { reb_error(sim, "REBOUNDx Error: Need to set speed of light in gr effect.  See examples in documentation.\n"); return false; }
        add(REBXAMD_OP_GR_FULL, -1, sim->gravity_ignore_terms == 1 ? 1 : 0, *c, 0.);
    } else if (fn == rebx_central_force) {
        for (const SourceC& s : b->central) add(REBXAMD_OP_CENTRAL, s.idx, 0, *s.A, *s.gamma);
    } else if (fn == rebx_tides_precession) {
        if (b->tides_primaries.empty()) add(REBXAMD_OP_TIDES, 0, 0, 0., 0.);
        for (int s : b->tides_primaries) add(REBXAMD_OP_TIDES, s, 0, 0., 0.);
    } else if (fn == rebx_modify_orbits_forces) {
        const int* cp = (const int*)rebx_get_param(rebx, f->ap, "coordinates");
        const int coords = cp ? *cp : REBX_COORDINATES_JACOBI;
        if (coords == REBX_COORDINATES_PARTICLE) {
            if (b->first_primary < 0) {
                reb_error(sim, "Coordinates set to REBX_COORDINATES_PARTICLE, but primary param was not found in any particle.  Need to set parameter.\n");
                return false;
            }
            add(REBXAMD_OP_MOF, b->first_primary, coords, 0., 0.);
        } else if (coords == REBX_COORDINATES_JACOBI || coords == REBX_COORDINATES_BARYCENTRIC) {
            add(REBXAMD_OP_MOF, -1, coords, 0., 0.);
        } else {
            reb_error(sim, "Coordinates not supported in REBOUNDx.\n");
            return false;
        }
    }
    return true;
}

// Host array-of-structs -> device columns: packs, re-flattens the parameter lists if they changed,
// uploads.  On success `v` describes the device copy; t[0..2] = times at start, after packing,
// after the upload.  A block of masses travels only when one of them differs from what the device
// holds.  with_acc = false: the accelerations stay on the host and the device columns start from
// zero (the caller adds what comes back), which saves their 24 bytes per particle on the way up.
// with_vel = false: the velocities stay behind too (the device columns keep whatever they held:
// for callers whose kernels do not read them).
// in_place (small systems only): nothing is uploaded -- the view addresses the pinned host columns themselves, which
// the kernels then read and update over the link (a few round trips instead of two copy commands and their latencies).
bool stage_in(struct reb_simulation* sim, struct rebx_extras* rebx, Bridge* b, const struct reb_particle* particles, const int N,
              rebxamd_view& v, double t[3], const bool with_acc = true, const bool with_vel = true, bool in_place = false) {
    if (!ensure_capacity(rebx, b, (size_t)N)) return false;
    in_place = in_place && small_system(b, N) && b->hd_dyn[0] != nullptr;
    t[0] = now_ms();
    double* const* h = b->h_dyn;
    hipStream_t st = nullptr;
    bool ok = true;
    // Large arrays go block by block: while the copy engine uploads one block of the columns, the host
    // threads pack the next one.
    const int block = small_system(b, N) ? N : pipeline_block(N);
    unsigned long long sum = 1469598103934665603ull;
    long long last_massive = -1;
    double pack_ms = 0.;
    for (int b0 = 0; b0 < N; b0 += block) {
        const int b1 = (N - b0 > block) ? b0 + block : N;
        const double tp = now_ms();
        const int nt = host_threads(b1 - b0);
        unsigned long long chunk_sum[64];
        long long chunk_last[64];
        bool chunk_mass[64];
        const size_t synced = b->mass_synced;
        parallel_chunks(b1 - b0, nt, [&](int lo, int hi, int k) {
            unsigned long long cs = 1469598103934665603ull;
            long long lm = -1;
            bool moved = false;
            for (int i = b0 + lo; i < b0 + hi; i++) {
                const struct reb_particle& p = particles[i];
                // (streaming stores: the staging columns are written once and read by the copy engine, not by this core)
                __builtin_nontemporal_store(p.x, &h[H_X][i]); __builtin_nontemporal_store(p.y, &h[H_Y][i]); __builtin_nontemporal_store(p.z, &h[H_Z][i]);
                if (with_vel) { __builtin_nontemporal_store(p.vx, &h[H_VX][i]); __builtin_nontemporal_store(p.vy, &h[H_VY][i]); __builtin_nontemporal_store(p.vz, &h[H_VZ][i]); }
                moved |= (size_t)i >= synced || bits_of(h[H_M][i]) != bits_of(p.m);
                h[H_M][i] = p.m;
                if (with_acc) { h[H_AX][i] = p.ax; h[H_AY][i] = p.ay; h[H_AZ][i] = p.az; }
                if (p.m != 0.) lm = i;
                cs = (cs ^ (unsigned long long)(uintptr_t)p.ap) * 1099511628211ull;
            }
            chunk_sum[k] = cs; chunk_last[k] = lm; chunk_mass[k] = moved;
        });
        // fold of the per-chunk folds, in order: changes whenever any list head changes
        sum = (sum ^ (unsigned long long)nt) * 1099511628211ull;
        bool mass_moved = false;
        for (int k = 0; k < nt; k++) {
            sum = (sum ^ chunk_sum[k]) * 1099511628211ull;
            if (chunk_last[k] > last_massive) last_massive = chunk_last[k];
            mass_moved |= chunk_mass[k];
        }
        pack_ms += now_ms() - tp;
        if (small_system(b, N)) {
            // (one block.)  In place only behind a short massive prefix, where one pass reads every column once; the passes
            // of the other mass layouts read them several times and are better served by device memory
            in_place = in_place && last_massive + 1 <= REBXAMD_MAX_HEAD;
            if (in_place) __builtin_ia32_sfence();      // (the streaming stores above must have left this core before a kernel reads them)
            else ok = hipMemcpyAsync(b->d_slab, b->h_slab, H_NDYN * b->cap * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess;
        } else for (int k = 0; k < H_NDYN && ok; k++) {
            if (k == H_M && !mass_moved) continue;
            if (k >= H_AX && !with_acc) continue;
            if (k >= H_VX && k <= H_VZ && !with_vel) continue;
            ok = hipMemcpyAsync(b->d_dyn[k] + b0, b->h_dyn[k] + b0, (size_t)(b1 - b0) * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess;
        }
    }
    if (!with_acc && !small_system(b, N))
        for (int k = H_AX; k <= H_AZ && ok; k++) ok = hipMemsetAsync(b->d_dyn[k], 0, (size_t)N * sizeof(double), st) == hipSuccess;
    b->mass_synced = (ok && !in_place) ? (size_t)N : 0;
    const bool rebuild = b->built_generation != b->generation || b->built_n != N || b->built_checksum != sum;
    const double tf = now_ms();
    bool changed[P_NPAR];
    if (rebuild) flatten_params(rebx, b, particles, N);
    else refresh_values(b, changed);        // the lists did not change; the values in them may have (src/core.c:614-622 hands out pointers)
    pack_ms += now_ms() - tf;
    t[1] = t[0] + pack_ms;              // time spent packing and flattening; the uploads ran underneath

    const size_t bytes = (size_t)N * sizeof(double);
    if (rebuild) {
        for (int k = 0; k < P_NPAR && ok; k++)
            if (b->par_used[k]) ok = hipMemcpyAsync(b->d_par[k], b->h_par[k], bytes, hipMemcpyHostToDevice, st) == hipSuccess;
        b->built_generation = b->generation; b->built_n = N; b->built_checksum = sum;
    } else {
        for (int k = 0; k < P_NPAR && ok; k++)
            if (changed[k]) ok = hipMemcpyAsync(b->d_par[k], b->h_par[k], bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    }
    // (small systems: the copy is left queued ahead of the kernels; the wait is not worth its latency)
    if (!small_system(b, N)) ok = ok && hipStreamSynchronize(st) == hipSuccess;
    t[2] = now_ms();
    if (!ok) { rebx_error(rebx, "REBOUNDx (MI355X) Error: host to device copy failed.\n"); return false; }

    memset(&v, 0, sizeof(v));
    v.n = N;
    v.n_head = last_massive + 1 < 1 ? 1 : last_massive + 1;
    v.tail_massless = 1;
    v.G = sim->G;
    double* const* cols = in_place ? b->hd_dyn : b->d_dyn;
    v.x = cols[H_X]; v.y = cols[H_Y]; v.z = cols[H_Z];
    v.vx = cols[H_VX]; v.vy = cols[H_VY]; v.vz = cols[H_VZ];
    v.m = cols[H_M];
    v.ax = cols[H_AX]; v.ay = cols[H_AY]; v.az = cols[H_AZ];
    v.beta = b->par_used[P_BETA] ? b->d_par[P_BETA] : nullptr;
    v.R_tides = b->par_used[P_RT] ? b->d_par[P_RT] : nullptr;
    v.k1 = b->par_used[P_K1] ? b->d_par[P_K1] : nullptr;
    v.tau_a = b->par_used[P_TA] ? b->d_par[P_TA] : nullptr;
    v.tau_e = b->par_used[P_TE] ? b->d_par[P_TE] : nullptr;
    v.tau_inc = b->par_used[P_TI] ? b->d_par[P_TI] : nullptr;
    return true;
}

// whether a call on N particles should try the in-place route (stage_in has the last word: it knows the masses)
bool in_place_wanted(const Bridge* b, int N) {
    static const bool allowed = [] { const char* e = getenv("REBXAMD_HOOK_IN_PLACE"); return !(e && e[0] == '0'); }();
    static const int limit = [] { const char* e = getenv("REBXAMD_HOOK_IN_PLACE_LIMIT"); return e ? atoi(e) : small_limit(); }();
    return allowed && b->in_place && N <= limit && small_system(b, N);
}

void report_warnings(struct reb_simulation* sim, uint32_t bits) {
    if (bits & WARN_GR_ITER)
        reb_warning(sim, "REBOUNDx Warning: 10 iterations in gr.c failed to converge. This is typically because the perturbation is too strong for the current implementation.");
    if (bits & WARN_GRFULL_SWEEPS) reb_warning(sim, "10 loops in rebx_gr_full did not converge.\n");
    if (bits & WARN_IM_ITER)
        reb_warning(sim, "REBOUNDx: 10 iterations in integrator_implicit_midpoint.c failed to converge. This is typically because the perturbation is too strong for the current implementation.");
}

// Runs a run of built-in forces on `particles` (host array-of-structs) in one device round trip.
void run_builtin(struct reb_simulation* sim, struct rebx_extras* rebx, const std::vector<Pending>& batch,
                 struct reb_particle* const particles, const int N) {
    if (batch.empty() || N <= 0) return;
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    // Large systems in the fast arithmetic leave the accelerations on the host: the device sums the forces from zero
    // and the sum is added here (one rounding of the total instead of one per force: below the fast mode's own bar).
    // The exact arithmetic adds force after force onto the uploaded values, as the reference does.
    // (gr_full reads the incoming accelerations -- src/gr_full.c:70-74 -- so a batch that holds it uploads them.)
    // A batch of position-only forces (the reference's REBX_FORCE_POS, src/core.c:232-263) leaves the velocities behind too.
    const bool large = !small_system(b, N);
    bool reads_acc = false, reads_vel = false;
    for (const Pending& pf : batch) {
        reads_acc |= pf.force->update_accelerations == rebx_gr_full;
        reads_vel |= pf.builtin->type != REBX_FORCE_POS;
    }
    const bool delta = b->mode == REBXAMD_MODE_FAST && !reads_acc && large;
    // Small systems without gr_full (whose all-pairs passes read every column many times) are evaluated where they are
    // packed: in the pinned host columns.  REBXAMD_HOOK_IN_PLACE=0 / rebxamd_set_in_place(rebx, 0) go back to one copy per direction.
    const bool in_place = in_place_wanted(b, N) && !reads_acc;
    if (!stage_in(sim, rebx, b, particles, N, v, t, !delta, reads_vel || !large, in_place)) return;
    const bool evaluated_in_place = in_place && v.x == b->hd_dyn[H_X];

    // ---- ops (effect parameters are read on every call: they are a handful of list walks)
    std::vector<rebxamd_op> ops;
    for (const Pending& pf : batch) ops_of_force(sim, rebx, b, pf, ops);
    if (ops.empty()) { b->timings[0] = t[1] - t[0]; b->timings[1] = t[2] - t[1]; b->timings[2] = b->timings[3] = 0.; return; }

    // ---- kernels
    hipStream_t st = nullptr;
    // (only gr and gr_full can raise a warning -- their iteration limits: a batch without them has no warning word to fetch)
    bool may_warn = false;
    for (const rebxamd_op& o : ops) may_warn |= o.kind == REBXAMD_OP_GR || o.kind == REBXAMD_OP_GR_FULL;
    const int rc = rebxamd_apply(b->ctx, &v, ops.data(), (int)ops.size(), b->mode, st);
    if (rc != REBXAMD_OK) {
        char str[400];
        snprintf(str, sizeof(str), "REBOUNDx (MI355X) Error: %s. Accelerations were not updated.\n", rebxamd_last_error(b->ctx));
        (void)hipStreamSynchronize(st);
        reb_error(sim, str);
        return;
    }
    bool ok = small_system(b, N) || hipStreamSynchronize(st) == hipSuccess;
    const double t3 = now_ms();

    // ---- device -> host, columns -> array-of-structs
    double* const* h = b->h_dyn;
    static_assert(H_AY == H_AX + 1 && H_AZ == H_AX + 2, "the acceleration columns are neighbours in the slab");
    uint32_t bits = 0;
    if (small_system(b, N)) {
        if (!evaluated_in_place) ok = hipMemcpyAsync(b->h_dyn[H_AX], b->d_dyn[H_AX], (2 * b->cap + (size_t)N) * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess;
        if (may_warn) ok = ok && rebxamd_fetch_warnings(b->ctx, &bits, st) == REBXAMD_OK;     // synchronises the stream
        else ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (!ok) { rebx_error(rebx, "REBOUNDx (MI355X) Error: device to host copy failed.\n"); return; }
        for (int i = 0; i < N; i++) { particles[i].ax = h[H_AX][i]; particles[i].ay = h[H_AY][i]; particles[i].az = h[H_AZ][i]; }
    } else {
        // block by block: the host threads scatter one block into the array of structs while the copy
        // engine downloads the next ones (one event per block marks its arrival)
        const int block = pipeline_block(N);
        const int nblocks = (N + block - 1) / block;
        while ((int)b->block_done.size() < nblocks && ok) {
            hipEvent_t e = nullptr;
            ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            if (ok) b->block_done.push_back(e);
        }
        for (int blk = 0; blk < nblocks && ok; blk++) {
            const int b0 = blk * block, b1 = (N - b0 > block) ? b0 + block : N;
            for (int k = H_AX; k <= H_AZ && ok; k++)
                ok = hipMemcpyAsync(b->h_dyn[k] + b0, b->d_dyn[k] + b0, (size_t)(b1 - b0) * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess;
            ok = ok && hipEventRecord(b->block_done[blk], st) == hipSuccess;
        }
        for (int blk = 0; blk < nblocks && ok; blk++) {
            const int b0 = blk * block, b1 = (N - b0 > block) ? b0 + block : N;
            ok = hipEventSynchronize(b->block_done[blk]) == hipSuccess;
            if (!ok) break;
            parallel_chunks(b1 - b0, host_threads(b1 - b0), [&](int lo, int hi, int) {
                if (delta) for (int i = b0 + lo; i < b0 + hi; i++) { particles[i].ax += h[H_AX][i]; particles[i].ay += h[H_AY][i]; particles[i].az += h[H_AZ][i]; }
                else for (int i = b0 + lo; i < b0 + hi; i++) { particles[i].ax = h[H_AX][i]; particles[i].ay = h[H_AY][i]; particles[i].az = h[H_AZ][i]; }
            });
        }
        ok = ok && rebxamd_fetch_warnings(b->ctx, &bits, st) == REBXAMD_OK;     // synchronises the stream
        if (!ok) { rebx_error(rebx, "REBOUNDx (MI355X) Error: device to host copy failed.\n"); return; }
    }
    const double t4 = now_ms();
    report_warnings(sim, bits);
    b->timings[0] = t[1] - t[0]; b->timings[1] = t[2] - t[1]; b->timings[2] = t3 - t[2]; b->timings[3] = t4 - t3;
}

void run_single(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) {
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    const Builtin* bi = builtin_of(force);
    if (!rebx || !bi) return;
    std::vector<Pending> one{{force, bi}};
    run_builtin(sim, rebx, one, particles, N);
}

}  // namespace

// reference src/core.c:868-880.  List order is the reverse of the order of rebx_add_force calls.
extern "C" void rebx_additional_forces(struct reb_simulation* sim) {
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    if (!rebx) return;
    const int N = sim->N - sim->N_var;
    std::vector<Pending> batch;
    for (struct rebx_node* cur = rebx->additional_forces; cur; cur = cur->next) {
        struct rebx_force* force = (struct rebx_force*)cur->object;
        if (const Builtin* bi = builtin_of(force)) {
            batch.push_back({force, bi});
        } else {
            // a user-supplied callback: everything before it must be on the host when it runs
            run_builtin(sim, rebx, batch, sim->particles, N);
            batch.clear();
            force->update_accelerations(sim, force, sim->particles, N);
        }
    }
    run_builtin(sim, rebx, batch, sim->particles, N);
}

// The force callbacks (reference src/core.h:66-74): one force on the array they are handed.
extern "C" void rebx_gr(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_gr_full(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_gr_potential(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_radiation_forces(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_modify_orbits_forces(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_tides_precession(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_central_force(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }
extern "C" void rebx_gravitational_harmonics(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N) { run_single(sim, force, particles, N); }

// ------------------------------------------------------------------ operators
static void free_operator(struct rebx_operator* op) {                               // src/core.c:780-786
    free(op->name);
    free_param_list(&op->ap);
    free(op);
}

extern "C" struct rebx_operator* rebx_create_operator(struct rebx_extras* const rebx, const char* name) {     // src/core.c:279-314
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return NULL; }
    struct rebx_operator* op = (struct rebx_operator*)checked_malloc(rebx, sizeof(*op));
    if (!op) return NULL;
    op->ap = NULL;
    op->sim = rebx->sim;
    op->operator_type = REBX_OPERATOR_NONE;
    op->step_function = NULL;
    op->name = NULL;
    if (name != NULL) {
        op->name = dup_name(rebx, name);
        if (!op->name) { free(op); return NULL; }
    }
    struct rebx_node* n = new_node(rebx, op);
    if (!n) { free(op->name); free(op); return NULL; }
    list_push_front(&rebx->allocated_operators, n);
    return op;
}

extern "C" struct rebx_operator* rebx_load_operator(struct rebx_extras* const rebx, const char* name) {       // src/core.c:316-368
    struct rebx_operator* op = rebx_create_operator(rebx, name);
    if (!op) return NULL;
    if (strcmp(name, "integrate_force") == 0) op->step_function = rebx_integrate_force;
    else if (strcmp(name, "modify_mass") == 0) op->step_function = rebx_modify_mass;
    else if (strcmp(name, "modify_orbits_direct") == 0) op->step_function = rebx_modify_orbits_direct;
    else if (strcmp(name, "track_min_distance") == 0) { op->step_function = rebx_track_min_distance; op->operator_type = REBX_OPERATOR_RECORDER; return op; }
    else if (strcmp(name, "drift") == 0) op->step_function = rebx_drift_step;
    else if (strcmp(name, "kick") == 0) op->step_function = rebx_kick_step;
    else {
        char str[300];
        snprintf(str, sizeof(str), "REBOUNDx error: Operator '%s' not found in REBOUNDx library.\n", name);
        rebx_error(rebx, str);
        rebx_remove_operator(rebx, op);
        return NULL;
    }
    op->operator_type = REBX_OPERATOR_UPDATER;
    return op;
}

extern "C" int rebx_add_operator_step(struct rebx_extras* rebx, struct rebx_operator* op, const double dt_fraction, enum rebx_timing timing) {   // src/core.c:411-460
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return 0; }
    if (op == NULL) { rebx_error(rebx, "REBOUNDx error: Passed NULL pointer to rebx_add_operator_step.\n"); return 0; }
    if (op->step_function == NULL) {
        rebx_error(rebx, "REBOUNDx error: Need to set step_function pointer on operator before adding to simulation. See custom effects example.\n");
        return 0;
    }
    if (op->operator_type == REBX_OPERATOR_NONE) {
        rebx_error(rebx, "REBOUNDx error: Need to set operator_type field on operator before adding to simulation. See custom effects example.\n");
        return 0;
    }
    if (timing != REBX_TIMING_PRE && timing != REBX_TIMING_POST) return 0;
    struct rebx_step* step = (struct rebx_step*)checked_malloc(rebx, sizeof(*step));
    if (!step) return 0;
    step->op = op;
    step->dt_fraction = dt_fraction;
    struct rebx_node* n = new_node(rebx, step);
    if (!n) { free(step); return 0; }
    struct reb_simulation* sim = rebx->sim;
    if (timing == REBX_TIMING_PRE) {
        list_push_front(&rebx->pre_timestep_modifications, n);
        if (sim->pre_timestep_modifications != NULL && sim->pre_timestep_modifications != rebx_pre_timestep_modifications)
            reb_warning(sim, "REBOUNDx Warning: pre_timestep_modifications was set in the simulation and is being overwritten by REBOUNDx. To incorporate both, you can add your own custom effects through REBOUNDx.\n");
        sim->pre_timestep_modifications = rebx_pre_timestep_modifications;
    } else {
        list_push_front(&rebx->post_timestep_modifications, n);
        if (sim->post_timestep_modifications != NULL && sim->post_timestep_modifications != rebx_post_timestep_modifications)
            reb_warning(sim, "REBOUNDx Warning: post_timestep_modifications was set in the simulation and is being overwritten by REBOUNDx. To incorporate both, you can add your own custom effects through REBOUNDx.\n");
        sim->post_timestep_modifications = rebx_post_timestep_modifications;
    }
    return 1;
}

extern "C" int rebx_add_operator(struct rebx_extras* rebx, struct rebx_operator* op) {      // src/core.c:462-506
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return 0; }
    if (op == NULL) { rebx_error(rebx, "REBOUNDx error: Passed NULL pointer to rebx_add_operator.\n"); return 0; }
    struct reb_simulation* const sim = rebx->sim;
    if (op->operator_type == REBX_OPERATOR_RECORDER) return rebx_add_operator_step(rebx, op, 1., REBX_TIMING_POST);
    switch (sim->integrator) {
    case REB_INTEGRATOR_IAS15:          // the step size is not known before the step: whole step afterwards
        return rebx_add_operator_step(rebx, op, 1., REBX_TIMING_POST);
    case REB_INTEGRATOR_WHFAST: {       // half a step before, half a step after
        const int before = rebx_add_operator_step(rebx, op, 1. / 2., REBX_TIMING_PRE);
        const int after = rebx_add_operator_step(rebx, op, 1. / 2., REBX_TIMING_POST);
        return before && after;
    }
    case REB_INTEGRATOR_MERCURIUS:
        if (op->operator_type == REBX_OPERATOR_UPDATER)
            reb_error(sim, "REBOUNDx Error: Operators that affect particle trajectories are not supported with Mercurius. Must add as forces.\n");
        return 0;
    default:
        return 0;
    }
}

// removes the first step that uses `op`; 1 if one was found (src/core.c:667-695)
static int remove_one_step(struct rebx_node** head, struct rebx_operator* op) {
    for (struct rebx_node** link = head; *link; link = &(*link)->next) {
        struct rebx_step* step = (struct rebx_step*)(*link)->object;
        if (step->op == op) {
            struct rebx_node* dead = *link;
            *link = dead->next;
            free(step);
            free(dead);
            return 1;
        }
    }
    return 0;
}

extern "C" int rebx_remove_operator(struct rebx_extras* rebx, struct rebx_operator* op) {   // src/core.c:697-728
    const int allocated = list_remove_object(&rebx->allocated_operators, op);
    int success = 0;
    while (remove_one_step(&rebx->pre_timestep_modifications, op)) success = 1;
    while (remove_one_step(&rebx->post_timestep_modifications, op)) success = 1;
    if (allocated) free_operator(op);
    return success;
}

extern "C" struct rebx_operator* rebx_get_operator(struct rebx_extras* const rebx, const char* const name) {  // src/core.c:637-648
    for (struct rebx_node* cur = rebx->allocated_operators; cur; cur = cur->next) {
        struct rebx_operator* op = (struct rebx_operator*)cur->object;
        if (op->name && strcmp(op->name, name) == 0) return op;
    }
    return NULL;
}

static void run_steps(struct reb_simulation* sim, struct rebx_node* head) {         // src/core.c:882-912
    const double dt = sim->dt;
    for (struct rebx_node* cur = head; cur; cur = cur->next) {
        struct rebx_step* step = (struct rebx_step*)cur->object;
#ifdef REBXAMD_WITH_REBOUND
        if (sim->integrator == REB_INTEGRATOR_IAS15 && sim->ri_ias15.epsilon != 0 && step->op->operator_type == REBX_OPERATOR_UPDATER)
            reb_warning(sim, "REBOUNDx: Operators that affect particle trajectories with adaptive timesteps can give spurious results. Use sim.ri_ias15.epsilon=0 for fixed timestep with IAS, or use a different integrator.");
#endif
        step->op->step_function(sim, step->op, dt * step->dt_fraction);
    }
}
extern "C" void rebx_pre_timestep_modifications(struct reb_simulation* sim) {
    if (struct rebx_extras* rebx = (struct rebx_extras*)sim->extras) run_steps(sim, rebx->pre_timestep_modifications);
}
extern "C" void rebx_post_timestep_modifications(struct reb_simulation* sim) {
    if (struct rebx_extras* rebx = (struct rebx_extras*)sim->extras) run_steps(sim, rebx->post_timestep_modifications);
}

extern "C" void rebx_drift_step(struct reb_simulation* const sim, struct rebx_operator* const, const double dt) {   // src/steppers.c:112-121
    struct reb_particle* const ps = sim->particles;
    for (int i = 0; i < sim->N; i++) {
        ps[i].x += dt * ps[i].vx; ps[i].y += dt * ps[i].vy; ps[i].z += dt * ps[i].vz;
    }
}

extern "C" void rebx_kick_step(struct reb_simulation* const sim, struct rebx_operator* const, const double dt) {    // src/steppers.c:123-133
#ifdef REBXAMD_WITH_REBOUND
    reb_update_acceleration(sim);
#endif
    struct reb_particle* const ps = sim->particles;
    for (int i = 0; i < sim->N; i++) {
        ps[i].vx += dt * ps[i].ax; ps[i].vy += dt * ps[i].ay; ps[i].vz += dt * ps[i].az;
    }
}

namespace {

typedef void (*force_fn)(struct reb_simulation* const, struct rebx_force* const, struct reb_particle* const, const int);

// The four schemes on host arrays, for a force that is a user callback (it can only run on the
// host).  Stage arrays are full copies of the particle array, as the callback contract expects
// (src/integrator_rk4.c:40-91, integrator_rk2.c:38-67, integrator_euler.c:33-41,
// integrator_implicit_midpoint.c:83-124).
void integrate_on_host(struct reb_simulation* sim, struct rebx_force* force, int integrator, double dt, int N) {
    struct reb_particle* const ps = sim->particles;
    const force_fn f = force->update_accelerations;
    auto stage = [&](std::vector<struct reb_particle>& k, const struct reb_particle* acc_of, double h) {
        for (int i = 0; i < N; i++) {
            k[i].vx = ps[i].vx + h * acc_of[i].ax; k[i].vy = ps[i].vy + h * acc_of[i].ay; k[i].vz = ps[i].vz + h * acc_of[i].az;
        }
    };
    if (integrator == REBXAMD_INTEGRATOR_EULER) {
        f(sim, force, ps, N);
        for (int i = 0; i < N; i++) { ps[i].vx += dt * ps[i].ax; ps[i].vy += dt * ps[i].ay; ps[i].vz += dt * ps[i].az; }
    } else if (integrator == REBXAMD_INTEGRATOR_RK2) {
        std::vector<struct reb_particle> k2(ps, ps + N);
        f(sim, force, ps, N);
        stage(k2, ps, 2. * dt / 3.);
        f(sim, force, k2.data(), N);
        const double b1 = dt / 4., b2 = 3. * dt / 4.;
        for (int i = 0; i < N; i++) {
            ps[i].vx += b1 * ps[i].ax + b2 * k2[i].ax; ps[i].vy += b1 * ps[i].ay + b2 * k2[i].ay; ps[i].vz += b1 * ps[i].az + b2 * k2[i].az;
        }
    } else if (integrator == REBXAMD_INTEGRATOR_RK4) {
        rebx_reset_accelerations(ps, N);
        std::vector<struct reb_particle> k2(ps, ps + N), k3(ps, ps + N);
        f(sim, force, ps, N);
        stage(k2, ps, dt / 2.);
        f(sim, force, k2.data(), N);
        stage(k3, k2.data(), dt / 2.);
        f(sim, force, k3.data(), N);
        stage(k2, k3.data(), dt);
        for (int i = 0; i < N; i++) { k3[i].ax += k2[i].ax; k3[i].ay += k2[i].ay; k3[i].az += k2[i].az; }
        rebx_reset_accelerations(k2.data(), N);
        f(sim, force, k2.data(), N);
        const double dt6 = dt / 6.;
        for (int i = 0; i < N; i++) {
            ps[i].vx += dt6 * (ps[i].ax + k2[i].ax + 2. * k3[i].ax);
            ps[i].vy += dt6 * (ps[i].ay + k2[i].ay + 2. * k3[i].ay);
            ps[i].vz += dt6 * (ps[i].az + k2[i].az + 2. * k3[i].az);
        }
    } else if (integrator == REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT) {
        std::vector<struct reb_particle> fin(ps, ps + N), avg(ps, ps + N), prev(N);
        int n = 0;
        for (n = 0; n < 10; n++) {
            prev = fin;
            f(sim, force, avg.data(), N);
            stage(fin, avg.data(), dt);
            double tot2 = 0., delta2 = 0.;
            for (int i = 0; i < N; i++) {
                const double dx = fin[i].vx - prev[i].vx, dy = fin[i].vy - prev[i].vy, dz = fin[i].vz - prev[i].vz;
                delta2 += dx * dx + dy * dy + dz * dz;
                tot2 += fin[i].vx * fin[i].vx + fin[i].vy * fin[i].vy + fin[i].vz * fin[i].vz;
            }
            if (delta2 / tot2 < DBL_EPSILON * DBL_EPSILON) break;
            for (int i = 0; i < N; i++) {
                avg[i].vx = 0.5 * (ps[i].vx + fin[i].vx); avg[i].vy = 0.5 * (ps[i].vy + fin[i].vy); avg[i].vz = 0.5 * (ps[i].vz + fin[i].vz);
                avg[i].ax = 0.; avg[i].ay = 0.; avg[i].az = 0.;
            }
        }
        if (n == 10) report_warnings(sim, WARN_IM_ITER);
        for (int i = 0; i < N; i++) { ps[i].vx = fin[i].vx; ps[i].vy = fin[i].vy; ps[i].vz = fin[i].vz; }
    }
}

}  // namespace

extern "C" void rebx_integrate_force(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt) {    // src/integrate_force.c:34-78
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    if (!rebx) return;
    struct rebx_force* force = (struct rebx_force*)rebx_get_param(rebx, op->ap, "force");
    if (force == NULL) {
        reb_error(sim, "REBOUNDx Error: Force parameter not set in rebx_integrate operator. See examples for how to add as a parameter.\n");
        return;                         // the reference goes on and dereferences the NULL force
    }
    int integrator = REBXAMD_INTEGRATOR_EULER;
    if (const int* ip = (const int*)rebx_get_param(rebx, op->ap, "integrator")) integrator = *ip;
    rebx_reset_accelerations(sim->particles, sim->N);
    const int N = sim->N - sim->N_var;
    if (N <= 0) return;
    if (integrator != REBXAMD_INTEGRATOR_EULER && integrator != REBXAMD_INTEGRATOR_RK2 && integrator != REBXAMD_INTEGRATOR_RK4 &&
        integrator != REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT)
        return;
    const Builtin* bi = builtin_of(force);
    if (!bi) {
        if (force->update_accelerations) integrate_on_host(sim, force, integrator, dt, N);
        return;
    }
    // built-in force: one round trip, every stage on the device
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    // (implicit midpoint makes a pass per round over the columns: beyond a wavefront's worth of bodies it keeps them in device
    // memory; gr_full reads them all the time)
    const bool in_place = in_place_wanted(b, N) && (integrator != REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT || N <= REBXAMD_MAX_HEAD) &&
                          force->update_accelerations != rebx_gr_full;
    if (!stage_in(sim, rebx, b, sim->particles, N, v, t, true, true, in_place)) return;
    const bool evaluated_in_place = in_place && v.x == b->hd_dyn[H_X];
    std::vector<rebxamd_op> ops;
    const Pending pf{force, bi};
    if (!ops_of_force(sim, rebx, b, pf, ops)) return;
    hipStream_t st = nullptr;
    // (the round count of the implicit scheme is not asked for: its warning travels in the warning word)
    const int rc = rebxamd_integrate_force(b->ctx, &v, ops.data(), (int)ops.size(), integrator, dt, b->mode, nullptr, st);
    if (rc != REBXAMD_OK) {
        char str[400];
        snprintf(str, sizeof(str), "REBOUNDx (MI355X) Error: %s. Velocities were not updated.\n", rebxamd_last_error(b->ctx));
        (void)hipStreamSynchronize(st);
        reb_error(sim, str);
        return;
    }
    bool ok = evaluated_in_place || hipStreamSynchronize(st) == hipSuccess;
    const double t3 = now_ms();
    const size_t bytes = (size_t)N * sizeof(double);
    static const int back[6] = {H_VX, H_VY, H_VZ, H_AX, H_AY, H_AZ};
    for (int k = 0; k < 6 && ok && !evaluated_in_place; k++) ok = hipMemcpyAsync(b->h_dyn[back[k]], b->d_dyn[back[k]], bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
    uint32_t bits = 0;
    ok = ok && rebxamd_fetch_warnings(b->ctx, &bits, st) == REBXAMD_OK;
    if (!ok) { rebx_error(rebx, "REBOUNDx (MI355X) Error: device to host copy failed.\n"); return; }
    double* const* h = b->h_dyn;
    struct reb_particle* const ps = sim->particles;
    for (int i = 0; i < N; i++) {
        ps[i].vx = h[H_VX][i]; ps[i].vy = h[H_VY][i]; ps[i].vz = h[H_VZ][i];
        ps[i].ax = h[H_AX][i]; ps[i].ay = h[H_AY][i]; ps[i].az = h[H_AZ][i];
    }
    const double t4 = now_ms();
    report_warnings(sim, bits);
    b->timings[0] = t[1] - t[0]; b->timings[1] = t[2] - t[1]; b->timings[2] = t3 - t[2]; b->timings[3] = t4 - t3;
}

// ------------------------------------------------------------------ potentials
namespace {
enum PotentialKind { POT_HARMONICS, POT_CENTRAL, POT_TIDES, POT_GR };

double host_potential(struct rebx_extras* rebx, PotentialKind kind, double c) {
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return 0; }
    struct reb_simulation* sim = rebx->sim;
    const int N = sim->N - sim->N_var;
    if (N <= 0) return 0.;
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    if (!stage_in(sim, rebx, b, sim->particles, N, v, t, true, true, in_place_wanted(b, N))) return 0.;
    std::vector<rebxamd_op> ops;
    auto add = [&](int k, int src, double a, double bb) {
        rebxamd_op o; o.kind = k; o.src = src; o.iarg = 0; o.reserved = 0; o.a = a; o.b = bb; ops.push_back(o);
    };
    if (kind == POT_HARMONICS) {
        for (const SourceJ& s : b->j2) add(REBXAMD_OP_J2, s.idx, *s.J, *s.Req);
        for (const SourceJ& s : b->j4) add(REBXAMD_OP_J4, s.idx, *s.J, *s.Req);
    } else if (kind == POT_CENTRAL) {
        for (const SourceC& s : b->central) add(REBXAMD_OP_CENTRAL, s.idx, *s.A, *s.gamma);
    } else if (kind == POT_TIDES) {
        // tides_precession.c:190-199: the LAST particle carrying "primary" wins (H is overwritten), else index 0
        int src = 0;
        for (int i = 0; i < N; i++)
            if (rebx_get_param(rebx, (struct rebx_node*)sim->particles[i].ap, "primary") != NULL) src = i;
        add(REBXAMD_OP_TIDES, src, 0., 0.);
    } else {
        add(REBXAMD_OP_GR_POTENTIAL, 0, c, 0.);
    }
    double H = 0.;
    for (size_t lo = 0; lo < ops.size(); lo += REBXAMD_MAX_OPS) {
        const int cnt = (int)std::min<size_t>(REBXAMD_MAX_OPS, ops.size() - lo);
        double out[REBXAMD_MAX_OPS];
        if (rebxamd_potentials(b->ctx, &v, ops.data() + lo, cnt, out, nullptr) != REBXAMD_OK) {
            char str[400];
            snprintf(str, sizeof(str), "REBOUNDx (MI355X) Error: %s.\n", rebxamd_last_error(b->ctx));
            reb_error(sim, str);
            return 0.;
        }
        for (int k = 0; k < cnt; k++) H += out[k];
    }
    return H;
}
}  // namespace

extern "C" double rebx_gravitational_harmonics_potential(struct rebx_extras* const rebx) { return host_potential(rebx, POT_HARMONICS, 0.); }
extern "C" double rebx_central_force_potential(struct rebx_extras* const rebx) { return host_potential(rebx, POT_CENTRAL, 0.); }
extern "C" double rebx_tides_precession_potential(struct rebx_extras* const rebx) { return host_potential(rebx, POT_TIDES, 0.); }
extern "C" double rebx_gr_potential_potential(struct rebx_extras* const rebx, const struct rebx_force* const gr_potential) {
    const double* c = (const double*)rebx_get_param(rebx, gr_potential->ap, "c");
    if (c == NULL) {
        rebx_error(rebx, "Need to set speed of light in gr effect.  See examples in documentation.\n");
        return 0;                       // the reference goes on and dereferences NULL
    }
    return host_potential(rebx, POT_GR, *c);
}

extern "C" double rebx_gr_hamiltonian(struct rebx_extras* const rebx, const struct rebx_force* const gr) {        // src/gr.c:248-260
    const double* c = (const double*)rebx_get_param(rebx, gr->ap, "c");
    if (c == NULL) { rebx_error(rebx, "Need to set speed of light in gr effect.  See examples in documentation.\n"); return 0; }
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return 0; }
    struct reb_simulation* sim = rebx->sim;
    const int N = sim->N - sim->N_var;
    if (N <= 0) return 0.;
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    if (!stage_in(sim, rebx, b, sim->particles, N, v, t, true, true, in_place_wanted(b, N))) return 0.;
    double H = 0.;
    if (rebxamd_gr_hamiltonian(b->ctx, &v, *c, &H, nullptr) != REBXAMD_OK) {
        char str[400];
        snprintf(str, sizeof(str), "REBOUNDx (MI355X) Error: %s.\n", rebxamd_last_error(b->ctx));
        reb_error(sim, str);
        return 0.;
    }
    return H;
}

// ------------------------------------------------------------------ operators that change the state
namespace {
// device -> host for the listed dynamic columns, then columns -> array-of-structs
// (in_place: the kernels updated the pinned host columns themselves; only the wait is left)
bool state_back(struct rebx_extras* rebx, Bridge* b, struct reb_particle* ps, int N, bool with_mass, bool in_place) {
    hipStream_t st = nullptr;
    const size_t bytes = (size_t)N * sizeof(double);
    bool ok = true;
    for (int k = H_X; k <= (with_mass ? H_M : H_VZ) && ok && !in_place; k++) ok = hipMemcpyAsync(b->h_dyn[k], b->d_dyn[k], bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    if (!ok) { rebx_error(rebx, "REBOUNDx (MI355X) Error: device to host copy failed.\n"); return false; }
    double* const* h = b->h_dyn;
    parallel_chunks(N, host_threads(N), [&](int lo, int hi, int) {
        for (int i = lo; i < hi; i++) {
            ps[i].x = h[H_X][i]; ps[i].y = h[H_Y][i]; ps[i].z = h[H_Z][i];
            ps[i].vx = h[H_VX][i]; ps[i].vy = h[H_VY][i]; ps[i].vz = h[H_VZ][i];
            if (with_mass) ps[i].m = h[H_M][i];
        }
    });
    return true;
}
// a parameter column nobody set was not uploaded: operators that take it unconditionally get a column of "not set"
const double* column_or_nan(Bridge* b, int k, int N) {
    if (!b->par_used[k]) {
        for (int i = 0; i < N; i++) b->h_par[k][i] = rebxamd::unset_value();
        (void)hipMemcpy(b->d_par[k], b->h_par[k], (size_t)N * sizeof(double), hipMemcpyHostToDevice);
    }
    return b->d_par[k];
}
void report_device_error(struct reb_simulation* sim, Bridge* b, const char* what) {
    char str[400];
    snprintf(str, sizeof(str), "REBOUNDx (MI355X) Error: %s. %s\n", rebxamd_last_error(b->ctx), what);
    (void)hipDeviceSynchronize();
    reb_error(sim, str);
}
}  // namespace

extern "C" void rebx_modify_mass(struct reb_simulation* const sim, struct rebx_operator* const, const double dt) {      // src/modify_mass.c:63-73
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    const int N = sim->N - sim->N_var;
    if (!rebx || N <= 0) return;
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    if (!stage_in(sim, rebx, b, sim->particles, N, v, t, true, true, in_place_wanted(b, N))) return;
    if (rebxamd_modify_mass(b->ctx, &v, column_or_nan(b, P_TM, N), dt, nullptr) != REBXAMD_OK) { report_device_error(sim, b, "Masses were not updated."); return; }
    state_back(rebx, b, sim->particles, N, true, v.x == b->hd_dyn[H_X]);
}

extern "C" void rebx_modify_orbits_direct(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt) {   // src/modify_orbits_direct.c:121-132
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    const int N = sim->N - sim->N_var;
    if (!rebx || N <= 0) return;
    const int* cp = (const int*)rebx_get_param(rebx, op->ap, "coordinates");
    const int coordinates = cp ? *cp : REBX_COORDINATES_JACOBI;
    const double* p_param = (const double*)rebx_get_param(rebx, op->ap, "p");
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    if (!stage_in(sim, rebx, b, sim->particles, N, v, t, true, true, in_place_wanted(b, N))) return;
    if (coordinates == REBX_COORDINATES_PARTICLE && b->first_primary < 0) {
        reb_error(sim, "Coordinates set to REBX_COORDINATES_PARTICLE, but primary param was not found in any particle.  Need to set parameter.\n");
        return;
    }
    if (coordinates != REBX_COORDINATES_JACOBI && coordinates != REBX_COORDINATES_BARYCENTRIC && coordinates != REBX_COORDINATES_PARTICLE) {
        reb_error(sim, "Coordinates not supported in REBOUNDx.\n");
        return;
    }
    if (rebxamd_modify_orbits_direct(b->ctx, &v, coordinates, b->first_primary, p_param, b->par_used[P_TOM] ? b->d_par[P_TOM] : nullptr,
                                     b->par_used[P_TON] ? b->d_par[P_TON] : nullptr, dt, nullptr) != REBXAMD_OK) {
        report_device_error(sim, b, "Orbits were not updated.");
        return;
    }
    state_back(rebx, b, sim->particles, N, false, v.x == b->hd_dyn[H_X]);
}

// reference src/core.h:74, src/ephemeris_forces.c: needs the JPL ephemeris readers and their data files, which are not on this
// path.  Exported so that bindings which resolve every force callback by name still load; it reports and changes nothing.
extern "C" void rebx_ephemeris_forces(struct reb_simulation* const sim, struct rebx_force* const, struct reb_particle* const, const int) {
    reb_error(sim, "REBOUNDx (MI355X) Error: ephemeris_forces is not part of this library (it needs the JPL ephemeris files). Accelerations were not updated.\n");
}

// reference src/track_min_distance.c:66-96.  Host code on the simulation's own particle array: a recorder reads a handful of
// particles and changes none.
extern "C" void rebx_track_min_distance(struct reb_simulation* const sim, struct rebx_operator* const, const double) {
    struct rebx_extras* rebx = (struct rebx_extras*)sim->extras;
    const int N = sim->N - sim->N_var;
    if (!rebx) return;
    for (int i = 0; i < N; i++) {
        struct reb_particle* const p = &sim->particles[i];
        if (!p->ap) continue;
        double* const min_distance = (double*)rebx_get_param(rebx, (struct rebx_node*)p->ap, "min_distance");
        if (!min_distance) continue;
        const uint32_t* const target = (const uint32_t*)rebx_get_param(rebx, (struct rebx_node*)p->ap, "min_distance_from");
        const struct reb_particle* source = &sim->particles[0];
        if (target) {
            source = nullptr;
            for (int j = 0; j < sim->N && !source; j++)
                if (sim->particles[j].hash == *target) source = &sim->particles[j];
            if (!source) { reb_error(sim, "REBOUNDx Error: min_distance_from names a particle hash that is not in the simulation.\n"); continue; }
        }
        const double dx = p->x - source->x, dy = p->y - source->y, dz = p->z - source->z;
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < *min_distance * (*min_distance)) {
            *min_distance = sqrt(r2);
            void* const orbit = rebx_get_param(rebx, (struct rebx_node*)p->ap, "min_distance_orbit");
            if (orbit) {
#ifdef REBXAMD_WITH_REBOUND
                *(struct reb_orbit*)orbit = reb_tools_particle_to_orbit(sim->G, *p, *source);
#else
                reb_warning(sim, "REBOUNDx (MI355X): min_distance_orbit is REBOUND's struct reb_orbit and is recorded only by a library built against REBOUND; min_distance was updated.");
#endif
            }
        }
    }
}

extern "C" double rebx_gr_full_hamiltonian(struct rebx_extras* const rebx, const struct rebx_force* const force) {      // src/gr_full.c:251-377
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return 0; }
    struct reb_simulation* sim = rebx->sim;
    const double* c = (const double*)rebx_get_param(rebx, force->ap, "c");
    if (c == NULL) { reb_error(sim, "Need to set speed of light in gr effect.  See examples in documentation.\n"); return 0; }
    const int N = sim->N - sim->N_var;
    if (N <= 0) return 0.;
    Bridge* b = bridge_of(rebx, true);
    rebxamd_view v;
    double t[3];
    if (!stage_in(sim, rebx, b, sim->particles, N, v, t)) return 0.;
    double H = 0.;
    if (rebxamd_gr_full_hamiltonian(b->ctx, &v, *c, &H, nullptr) != REBXAMD_OK) { report_device_error(sim, b, ""); return 0.; }
    return H;
}

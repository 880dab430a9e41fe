// pm_api.hip -- host side of libraytracer_amd.so: context, device buffers, kernel launches and the
// extern "C" entry points declared in include/raytracer_amd.h.  This is the layer that stands in
// for the reference's JobSystem (src/JobSystem.cpp): instead of queueing one job per image row for
// CPU workers, a frame is a short chain of kernels on one HIP stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "pm_build.h"
#include "pm_kernels.h"
#include "pm_frontend.h"

#include "pm_sort.h"
using namespace pm;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(PM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)
#define PMCHK(expr) do { int r_ = (expr); if (r_ != PM_OK) return r_; } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PM_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return fail(PM_EHIP, "hipFree failed"); }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(PM_ENOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e)); }
        cap = want;
        return PM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- scratch of pm_emit_range_static, the one entry point that allocates nothing: a file-scope device array (every GPU
// the library runs on holds its own copy until the process ends).  Host code never uses the array's name as an address:
// pm_ctx_create asks the runtime for the device address (hipGetSymbolAddress, with that device current) and keeps it
// in the context; kernels get that pointer as an argument, exactly as they get the slot buffers in pm_emit_range.
constexpr uint64_t ES_MAX = PM_EMIT_STATIC_MAX_ATTEMPTS;
constexpr size_t es_align(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr size_t ES_OFF_TMP = 0;                                                   // ES_MAX records of 32 bytes (k_emit)
constexpr size_t ES_OFF_FLAG = ES_OFF_TMP + es_align(ES_MAX * sizeof(pm_photon));  // one byte per attempt
constexpr size_t ES_OFF_COUNTS = ES_OFF_FLAG + es_align(ES_MAX);                   // successes per block of 256 attempts
constexpr size_t ES_OFF_OFFS = ES_OFF_COUNTS + es_align((ES_MAX / 256) * 4);       // their exclusive scan + the total
constexpr size_t ES_BYTES = ES_OFF_OFFS + es_align((ES_MAX / 256 + 2) * 8);        // + the work counter of k_emit_stream
static __device__ __align__(256) unsigned char scratch[ES_BYTES];

constexpr size_t LDS_SCENE_MAX = 48 * 1024;   // scenes up to this size are staged in LDS; larger ones are read from global memory
constexpr int N_EVENTS = 8;
constexpr int N_SCHED = 20;        // gather launches of a frame, at most: per map 4 group parts + 4 per-query launches behind them + 1 beside them

struct BuildSlot {
    hipStream_t stream = nullptr;
    DevBuf emit_tmp, emit_flag, emit_counts, emit_offs, emit_run, keys_in, keys_out, vals_in, vals_out, sort_tmp, bounds;
    void release() {
        DevBuf *b[] = {&emit_tmp, &emit_flag, &emit_counts, &emit_offs, &emit_run, &keys_in, &keys_out, &vals_in, &vals_out, &sort_tmp, &bounds};
        for (DevBuf *x : b) x->release();
    }
};

struct pm_ctx {
    int device = 0;
    unsigned char *static_scratch = nullptr;   // device address of `scratch` on this context's GPU (hipGetSymbolAddress)
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;           // the caustic gather runs beside the global one
    hipStream_t aux2 = nullptr;          // and the per-query gather of the bounced pixels' shade points beside both
    hipEvent_t ev_join2 = nullptr;
    hipStream_t sp_stream = nullptr;     // the ordering of the bounced pixels' shade points, beside the shadow pass (PM_SP_STREAM=0: behind it, on the caller's stream)
    hipEvent_t ev_sec = nullptr, ev_sp = nullptr;
    bool sp_on_side = true;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_g[2][3] = {};          // per map: gather begins, group kernel done, per-query kernel done
    hipEvent_t ev_part[2][4] = {};       // per map: a part of the group launches is done (its leftovers may be gathered)
    int tile_parts = 1;                  // launches the groups of a map are cut into (PM_TILE_PARTS, 1..4; measured: more than one is slower)
    // scene
    int nprims = 0, stride = 0;
    double yield[2] = {0.60, 0.36};   // stored photons per emission attempt (global, caustic): the reference scene's to begin with, then
                                      // what the last map of this scene showed (sizes the one emission range a map is made from)
    double loose_tiles = 8.0;         // a pixel tile deeper than this many tile spans is not a gather group (PM_LOOSE_TILES; 0: every tile is one)
    bool loose_auto = true;           // ... decided from the frame before: on when many tile groups were handed over as too spread out
    bool loose_on = false;            //     (many small objects), off again when few tiles are that deep; PM_LOOSE_TILES set: always as told
    bool emit_run = true;             // scenes without a sphere grid: lanes refilled within a wave's run of attempts (PM_EMIT_RUN=0: one attempt per lane, k_emit)
    bool emit_stream = true;          // sphere grid: emission as a stream of attempts per wave (PM_EMIT_STREAM=0: one attempt per lane, for A/B runs)
    int emit_blocks = 384;            // blocks of k_emit_stream per launch: two maps side by side take three quarters of 256 CUs x 4 blocks and leave
                                      // room for the frame's first passes (PM_EMIT_BLOCKS; 320 ... 448: 7.84 ms per step on 10 000 spheres, 512 / 640: 7.95)
    bool shadow_queue = true;         // sphere grid: the rays of a wave share a queue of tests (PM_SHADOW_QUEUE=0: every ray tests on its own, for A/B runs)
    bool shadow_per_ray = true;       // one lane per shadow ray rather than per shade point (PM_SHADOW_PER_RAY=0: the latter, for A/B runs)
    int scene_mode = SCENE_LDS;   // SCENE_LDS: staged per block; SCENE_GLOBAL: read in place; SCENE_GRID: in place, through the sphere grid
    size_t scene_b = 0;
    DevBuf scene, sgrid;
    SceneView gv{};
    // frame work buffers (grow only, reused across frames)
    DevBuf geo, obj, pixel, node, tabslot, direct, radg, radc, color, clist, tab, ctr, pane, sample, sched, groups, left, strips;
    DevBuf sp_keys_in, sp_keys_out, sp_vals_in, sp_vals_out, sp_sort_tmp, sp_bounds, sp_groups, sp_early, sp_pieces, sp_loose;   // spatial order of the bounced pixels' shade points
    bool use_tiles = true;     // PM_GATHER_TILE=0: every shade point goes through the per-query kernel (A/B switch)
    double sp_cell = 6.0;      // cell size of those groups in 63-NN radii (PM_SP_CELL), and the shortest run worth a group (PM_SP_MIN_RUN)
    int sp_min_run = 16;
    double sp_est = 1.0;       // a piece is grouped when its expected candidate set is within this many times PM_TILE_EST_MAX (PM_SP_EST)
    bool group_bounced = true; // PM_GROUP_BOUNCED=0: the bounced pixels' shade points stay with the per-query kernel (A/B switch)
    int tile_blocks = 0, query_blocks = 0, pair_blocks = 0;   // resident blocks of the gather kernels on this device
    bool use_pairs = true;     // PM_PAIR=0: the per-query gather takes one shade point per wave (k_gather) instead of two (k_gather_pair)
    bool debug = false;        // PM_DEBUG in the environment when the context was made, or pm_ctx_set_option("debug"): per-frame statistics on stderr
    long sp_cap_limit = 0;     // pm_ctx_set_option("sp_cap_limit") (tests): pretend the shade point queue is this small
    hipEvent_t ev_frame = nullptr;   // the end of the latest frame on the caller's stream, behind the joins with the other two
    unsigned long long frame_seq = 0, frame_waited = 0;   // frames queued; the latest one a host wait has covered
    uint32_t sp_cap = 0, c_cap = 0, b_cap = 0;   // slots of the shade point queue; node tables (bounced pixels); bounced pixels' shade points the spatial sort takes
    bool yield_known[2] = {false, false};        // a map of this scene and kind has reported what an emission attempt yields
    Counters *h_ctr = nullptr;   // pinned
    uint64_t *h_u64 = nullptr;   // pinned scratch (2 words)
    // emission / build scratch: slots 0 and 1 have streams of their own (pm_map_create / pm_map_build leave the work there
    // and return; two maps are built side by side, and beside the first passes of the frame that needs them); slot 2 runs
    // on the caller's stream (pm_emit_range, and whatever has to be redone synchronously)
    BuildSlot slot[3];
    int next_slot = 0;
    hipEvent_t ev_order = nullptr;       // "the caller's stream got here": a build starts behind it
    hipEvent_t ev[N_EVENTS] = {};
    std::vector<struct pm_map *> spare;   // destroyed maps whose device buffers wait for the next map (a frame rebuilds both maps)
};

struct pm_map {
    pm_ctx *ctx = nullptr;
    long long n = 0;
    // a map whose emission + build are still on a slot's stream: `ready` marks their end, *h_got (pinned) then holds the
    // photons the one emission range produced; map_ready() waits, checks, and emits more in the rare case it fell short
    bool pending = false;
    int failed = PM_OK;                  // sticky: a map whose emission / assembly / build failed reports that on every later use
    std::string fail_msg;
    int slot = 2;
    hipEvent_t ready = nullptr;
    hipEvent_t t_begin = nullptr, t_emitted = nullptr;   // on the build stream: emission starts / is over (the index build follows up to `ready`)
    uint64_t *h_got = nullptr;           // pinned, two words: photons available; a shard overflowed
    unsigned long long used_in_frame = 0;   // the latest frame whose gathers read this map (pm_ctx::frame_seq); pm_map_destroy waits for it
    int kind = 0; uint64_t seed = 0, next_attempt = 0; double lc[3] = {0, 0, 0};
    DevBuf records, pos4, rest, cells, keys, cbox, grid;
    MapView view{};
};

static void free_map(pm_map *map);
static int set_device(pm_ctx *ctx) { HIPCHK(hipSetDevice(ctx->device)); return PM_OK; }

extern "C" const char *pm_last_error(void) { return g_err.c_str(); }

static int ctx_init(pm_ctx *ctx) {
    HIPCHK(hipHostMalloc((void **)&ctx->h_ctr, sizeof(Counters), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&ctx->h_u64, 4 * sizeof(uint64_t), hipHostMallocDefault));
    for (int i = 0; i < N_EVENTS; i++) HIPCHK(hipEventCreate(&ctx->ev[i]));
    // two streams beside the caller's: more would share hardware queues (4 per process) and serialise behind each other.
    // They carry the map builds, and later in the frame the caustic gather and the bounced pixels' gathers.
    for (int i = 0; i < 2; i++) HIPCHK(hipStreamCreateWithFlags(&ctx->slot[i].stream, hipStreamNonBlocking));
    ctx->aux = ctx->slot[0].stream; ctx->aux2 = ctx->slot[1].stream;
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_frame, hipEventDisableTiming));
    ctx->debug = getenv("PM_DEBUG") != nullptr;      // (the environment is read here, once, not per frame)
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithFlags(&ctx->sp_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_sec, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_sp, hipEventDisableTiming));
    if (const char *e = getenv("PM_SP_STREAM")) ctx->sp_on_side = e[0] != '0';
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) for (int i = 0; i < 3; i++) HIPCHK(hipEventCreate(&ctx->ev_g[k][i]));
    for (int k = 0; k < 2; k++) for (int i = 0; i < 4; i++) HIPCHK(hipEventCreateWithFlags(&ctx->ev_part[k][i], hipEventDisableTiming));
    // this GPU's copy of the static scratch array (the device is current: pm_ctx_create set it)
    HIPCHK(hipGetSymbolAddress((void **)&ctx->static_scratch, HIP_SYMBOL(scratch)));
    if (!ctx->static_scratch) return fail(PM_EHIP, "hipGetSymbolAddress(scratch) returned NULL");
    return PM_OK;
}

extern "C" int pm_ctx_create(int device, pm_ctx **out) {
    if (!out) return fail(PM_EINVAL, "pm_ctx_create: out is NULL");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PM_EINVAL, "pm_ctx_create: no such device " + std::to_string(device));
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(PM_ESTATE, std::string("pm_ctx_create: this library is built for gfx950 (MI355X) only, device is ") + prop.gcnArchName);
    pm_ctx *ctx = new (std::nothrow) pm_ctx();
    if (!ctx) return fail(PM_ENOMEM, "out of host memory");
    ctx->device = device;
    int rc = ctx_init(ctx);
    if (rc != PM_OK) { std::string msg = g_err; pm_ctx_destroy(ctx); g_err = msg; return rc; }   // nothing of a half-made context is left behind
    *out = ctx;
    return PM_OK;
}

extern "C" int pm_ctx_destroy(pm_ctx *ctx) {
    if (!ctx) return PM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; i++) if (ctx->slot[i].stream) { (void)hipStreamSynchronize(ctx->slot[i].stream); (void)hipStreamDestroy(ctx->slot[i].stream); }
    for (int i = 0; i < 3; i++) ctx->slot[i].release();
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->ev_frame) (void)hipEventDestroy(ctx->ev_frame);
    DevBuf *bufs[] = {&ctx->scene, &ctx->sgrid, &ctx->geo, &ctx->obj, &ctx->pixel, &ctx->node, &ctx->tabslot, &ctx->direct, &ctx->radg, &ctx->radc,
                      &ctx->color, &ctx->clist, &ctx->tab, &ctx->ctr, &ctx->pane, &ctx->sample, &ctx->sched, &ctx->groups, &ctx->left, &ctx->strips,
                      &ctx->sp_keys_in, &ctx->sp_keys_out, &ctx->sp_vals_in, &ctx->sp_vals_out, &ctx->sp_sort_tmp, &ctx->sp_bounds, &ctx->sp_groups, &ctx->sp_early, &ctx->sp_pieces, &ctx->sp_loose};
    for (DevBuf *b : bufs) b->release();
    for (pm_map *m : ctx->spare) free_map(m);
    ctx->spare.clear();
    if (ctx->h_ctr) (void)hipHostFree(ctx->h_ctr);
    if (ctx->h_u64) (void)hipHostFree(ctx->h_u64);
    for (int i = 0; i < N_EVENTS; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    if (ctx->sp_stream) { (void)hipStreamSynchronize(ctx->sp_stream); (void)hipStreamDestroy(ctx->sp_stream); }
    if (ctx->ev_sec) (void)hipEventDestroy(ctx->ev_sec);
    if (ctx->ev_sp) (void)hipEventDestroy(ctx->ev_sp);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (int k = 0; k < 2; k++) for (int i = 0; i < 3; i++) if (ctx->ev_g[k][i]) (void)hipEventDestroy(ctx->ev_g[k][i]);
    for (int k = 0; k < 2; k++) for (int i = 0; i < 4; i++) if (ctx->ev_part[k][i]) (void)hipEventDestroy(ctx->ev_part[k][i]);
    delete ctx;
    return PM_OK;
}

extern "C" int pm_ctx_synchronize(pm_ctx *ctx) {
    if (!ctx) return fail(PM_EINVAL, "ctx is NULL");
    PMCHK(set_device(ctx));
    for (int i = 0; i < 2; i++) if (ctx->slot[i].stream) HIPCHK(hipStreamSynchronize(ctx->slot[i].stream));
    if (ctx->sp_stream) HIPCHK(hipStreamSynchronize(ctx->sp_stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}
extern "C" int pm_ctx_set_option(pm_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return fail(PM_EINVAL, "pm_ctx_set_option: NULL argument");
    const std::string n(name);
    if (n == "debug") ctx->debug = value != 0;
    else if (n == "sp_cap_limit") { if (value < 0 || value > 4e9) return fail(PM_EINVAL, "pm_ctx_set_option: sp_cap_limit out of range"); ctx->sp_cap_limit = (long)value; }
    else return fail(PM_EINVAL, "pm_ctx_set_option: unknown option '" + n + "' (debug, sp_cap_limit)");
    return PM_OK;
}
extern "C" int pm_ctx_set_stream(pm_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(PM_EINVAL, "ctx is NULL");
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return PM_OK;
}

// ------------------------------------------------------------------------------------ scene grid
// Host-side build of the uniform grid over spheres (the scene is uploaded once, before any timed region).
constexpr int SCENE_GRID_MIN_SPHERES = 64;     // fewer: the linear scan (from LDS when the scene fits) is as good
constexpr int SCENE_GRID_MAX_RES = 256;        // cells per axis
constexpr double SCENE_GRID_CELLS_PER_SPHERE = 4.0;
constexpr long long SCENE_GRID_MAX_CELLS_PER_SPHERE = 512;   // a sphere touching more cells is tested with the planes
struct HostSceneGrid {
    std::vector<int> loose, refs;
    std::vector<uint32_t> cell_start;
    int res[3] = {1, 1, 1};
    double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1}, cs[3] = {1, 1, 1};
};
static void build_scene_grid(const pm_prim *prims, int n, HostSceneGrid &hg) {
    // box of everywhere the sphere can be for dt in [0,1] (SphereObject::intersect moves the centre by dt * velocity,
    // src/Scene.cpp:90-92), padded far beyond the rounding of the walk
    struct Box { double lo[3], hi[3]; bool ok; };
    std::vector<Box> box(n);
    double glo[3] = {1e300, 1e300, 1e300}, ghi[3] = {-1e300, -1e300, -1e300};
    int ns = 0;
    for (int i = 0; i < n; i++) {
        const pm_prim &p = prims[i];
        Box &b = box[i];
        b.ok = false;
        if (p.kind != 0) continue;
        const double c[3] = {p.x, p.y, p.z}, v[3] = {p.dx, p.dy, p.dz};
        double r = fabs(p.radius);
        bool finite = std::isfinite(r);
        for (int a = 0; a < 3; a++) finite = finite && std::isfinite(c[a]) && std::isfinite(v[a]) && fabs(c[a]) < 1e12 && fabs(v[a]) < 1e12;
        if (!finite || r > 1e12) continue;
        for (int a = 0; a < 3; a++) {
            double c1 = c[a] + v[a];
            double pad = 1e-6 * (r + fabs(v[a])) + 1e-9 * (1.0 + fabs(c[a]) + fabs(c1));
            b.lo[a] = std::min(c[a], c1) - r - pad;
            b.hi[a] = std::max(c[a], c1) + r + pad;
            glo[a] = std::min(glo[a], b.lo[a]); ghi[a] = std::max(ghi[a], b.hi[a]);
        }
        b.ok = true;
        ns++;
    }
    for (int i = 0; i < n; i++) if (!box[i].ok) hg.loose.push_back(i);     // planes, and spheres that are not finite
    if (ns == 0) { hg.cell_start.assign(2, 0); return; }
    double ext[3], vol = 1;
    for (int a = 0; a < 3; a++) {
        double pad = 1e-6 * (ghi[a] - glo[a]) + 1e-9;
        glo[a] -= pad; ghi[a] += pad;
        ext[a] = ghi[a] - glo[a];
        vol *= ext[a];
    }
    double density = SCENE_GRID_CELLS_PER_SPHERE;
    if (const char *e = getenv("PM_SCENE_GRID_DENSITY")) { double v = atof(e); if (v > 0.01 && v < 1000) density = v; }   // experiments
    double cell = cbrt(vol / (density * ns));
    for (int a = 0; a < 3; a++) {
        int r = (int)ceil(ext[a] / cell);
        hg.res[a] = std::max(1, std::min(SCENE_GRID_MAX_RES, r));
        hg.lo[a] = glo[a]; hg.hi[a] = ghi[a];
        hg.cs[a] = ext[a] / hg.res[a];
    }
    auto cell_range = [&](const Box &b, int a, int &i0, int &i1) {
        i0 = (int)floor((b.lo[a] - hg.lo[a]) / hg.cs[a]); i1 = (int)floor((b.hi[a] - hg.lo[a]) / hg.cs[a]);
        i0 = std::max(0, std::min(hg.res[a] - 1, i0)); i1 = std::max(0, std::min(hg.res[a] - 1, i1));
    };
    size_t ncell = (size_t)hg.res[0] * hg.res[1] * hg.res[2];
    std::vector<uint32_t> count(ncell + 1, 0);
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) {
            Box &b = box[i];
            if (!b.ok) continue;
            int x0, x1, y0, y1, z0, z1;
            cell_range(b, 0, x0, x1); cell_range(b, 1, y0, y1); cell_range(b, 2, z0, z1);
            if (pass == 0 && (long long)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > SCENE_GRID_MAX_CELLS_PER_SPHERE) {
                b.ok = false; hg.loose.push_back(i);
                continue;
            }
            for (int z = z0; z <= z1; z++) for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++) {
                size_t c = ((size_t)z * hg.res[1] + y) * hg.res[0] + x;
                if (pass == 0) count[c]++;
                else hg.refs[count[c]++] = i;          // spheres arrive in index order: every cell's list is ascending
            }
        }
        if (pass == 0) {
            hg.cell_start.resize(ncell + 1);
            uint32_t run = 0;
            for (size_t c = 0; c < ncell; c++) { hg.cell_start[c] = run; run += count[c]; count[c] = hg.cell_start[c]; }
            hg.cell_start[ncell] = run;
            hg.refs.resize(run);
        }
    }
    std::sort(hg.loose.begin(), hg.loose.end());
}

// ---------------------------------------------------------------------------------------- scene
extern "C" int pm_scene_upload(pm_ctx *ctx, const pm_prim *prims, int32_t n) {
    if (!ctx || !prims || n <= 0) return fail(PM_EINVAL, "pm_scene_upload: need a context and n > 0 primitives");
    if (n > (1 << 23)) return fail(PM_EINVAL, "pm_scene_upload: at most 2^23 primitives (object id is packed into 24 bits)");
    PMCHK(set_device(ctx));
    int stride = (n + 3) & ~3;
    size_t bytes = scene_bytes(stride);
    std::vector<unsigned char> host(bytes, 0);
    double *g = reinterpret_cast<double *>(host.data());
    float *rgb = reinterpret_cast<float *>(host.data() + (size_t)stride * G_NFIELDS * sizeof(double));
    int *kind = reinterpret_cast<int *>(host.data() + (size_t)stride * (G_NFIELDS * sizeof(double) + 3 * sizeof(float)));
    for (int i = 0; i < n; i++) {
        const pm_prim &p = prims[i];
        if (p.kind != 0 && p.kind != 1) return fail(PM_EINVAL, "pm_scene_upload: prim " + std::to_string(i) + " has unknown kind");
        g[G_X * stride + i] = p.x; g[G_Y * stride + i] = p.y; g[G_Z * stride + i] = p.z;
        g[G_RADIUS * stride + i] = p.radius;
        g[G_DX * stride + i] = p.dx; g[G_DY * stride + i] = p.dy; g[G_DZ * stride + i] = p.dz;
        g[G_MIN * stride + i] = p.min_bound; g[G_MAX * stride + i] = p.max_bound;
        g[G_ABSORB * stride + i] = p.absorb_chance; g[G_DIFFUSE * stride + i] = p.diffuse_chance;
        g[G_SPECULAR * stride + i] = p.specular_chance; g[G_TRANSMISSION * stride + i] = p.transmission_chance;
        g[G_REFRACTION * stride + i] = p.refraction;
        rgb[0 * stride + i] = p.red; rgb[1 * stride + i] = p.green; rgb[2 * stride + i] = p.blue;
        // PlaneObject::intersect picks its branch from the field values (src/Scene.cpp:136,157,174)
        if (p.kind == 0) kind[i] = PK_SPHERE;
        else kind[i] = p.x != 0 ? PK_PLANE_X : (p.y != 0 ? PK_PLANE_Y : (p.z > 0 ? PK_PLANE_Z : PK_NEVER));
    }
    for (int i = n; i < stride; i++) kind[i] = PK_NEVER;
    PMCHK(ctx->scene.ensure(bytes));
    HIPCHK(hipMemcpyAsync(ctx->scene.p, host.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->nprims = n; ctx->stride = stride; ctx->scene_b = bytes;
    ctx->yield[0] = 0.60; ctx->yield[1] = 0.36;
    ctx->yield_known[0] = ctx->yield_known[1] = false;
    ctx->loose_on = false;
    ctx->scene_mode = bytes <= LDS_SCENE_MAX ? SCENE_LDS : SCENE_GLOBAL;
    unsigned char *d = ctx->scene.as<unsigned char>();
    ctx->gv = SceneView{};
    ctx->gv.n = n; ctx->gv.stride = stride;
    ctx->gv.g = reinterpret_cast<const double *>(d);
    ctx->gv.rgb = reinterpret_cast<const float *>(d + (size_t)stride * G_NFIELDS * sizeof(double));
    ctx->gv.kind = reinterpret_cast<const int *>(d + (size_t)stride * (G_NFIELDS * sizeof(double) + 3 * sizeof(float)));
    // many spheres: a uniform grid narrows down the primitives a ray is tested against (PM_SCENE_GRID=0: never, for A/B runs)
    const char *sg = getenv("PM_SCENE_GRID");
    int nspheres = 0;
    for (int i = 0; i < n; i++) nspheres += prims[i].kind == 0;
    if (nspheres >= SCENE_GRID_MIN_SPHERES && !(sg && sg[0] == '0')) {
        HostSceneGrid hg;
        build_scene_grid(prims, n, hg);
        size_t b_loose = (hg.loose.size() * sizeof(int) + 255) & ~(size_t)255, b_start = (hg.cell_start.size() * sizeof(uint32_t) + 255) & ~(size_t)255;
        size_t b_refs = (hg.refs.size() * sizeof(int) + 255) & ~(size_t)255, b_packed = hg.refs.size() * 8 * sizeof(double);
        std::vector<double> packed(hg.refs.size() * 8);
        for (size_t r = 0; r < hg.refs.size(); r++) {
            const pm_prim &p = prims[hg.refs[r]];
            double *P = &packed[r * 8];
            P[0] = p.x; P[1] = p.y; P[2] = p.z; P[3] = p.radius; P[4] = p.dx; P[5] = p.dy; P[6] = p.dz;
            long long id = hg.refs[r];
            memcpy(&P[7], &id, 8);
        }
        PMCHK(ctx->sgrid.ensure(b_loose + b_start + b_refs + b_packed + 256));
        unsigned char *gd = ctx->sgrid.as<unsigned char>();
        if (!hg.loose.empty()) HIPCHK(hipMemcpyAsync(gd, hg.loose.data(), hg.loose.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(gd + b_loose, hg.cell_start.data(), hg.cell_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        if (!hg.refs.empty()) {
            HIPCHK(hipMemcpyAsync(gd + b_loose + b_start, hg.refs.data(), hg.refs.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(gd + b_loose + b_start + b_refs, packed.data(), b_packed, hipMemcpyHostToDevice, ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        SceneGrid &G = ctx->gv.grid;
        G.nloose = (int)hg.loose.size(); G.rx = hg.res[0]; G.ry = hg.res[1]; G.rz = hg.res[2];
        G.loose = reinterpret_cast<const int *>(gd);
        G.cell_start = reinterpret_cast<const uint32_t *>(gd + b_loose);
        G.refs = reinterpret_cast<const int *>(gd + b_loose + b_start);
        G.packed = reinterpret_cast<const double *>(gd + b_loose + b_start + b_refs);
        for (int a = 0; a < 3; a++) { G.lo[a] = hg.lo[a]; G.hi[a] = hg.hi[a]; G.cs[a] = hg.cs[a]; G.inv_cs[a] = 1.0 / hg.cs[a]; }
        ctx->scene_mode = SCENE_GRID;
        if (ctx->debug) fprintf(stderr, "[pm] scene grid %d x %d x %d, %zu refs for %d spheres, %zu loose primitives\n", G.rx, G.ry, G.rz,
                                        hg.refs.size(), nspheres, hg.loose.size());
    }
    if (const char *e = getenv("PM_SHADOW_PER_RAY")) ctx->shadow_per_ray = e[0] != '0';
    if (const char *e = getenv("PM_SHADOW_QUEUE")) ctx->shadow_queue = e[0] != '0';
    if (const char *e = getenv("PM_EMIT_STREAM")) ctx->emit_stream = e[0] != '0';
    if (const char *e = getenv("PM_EMIT_RUN")) ctx->emit_run = e[0] != '0';
    if (const char *e = getenv("PM_EMIT_BLOCKS")) { int v = atoi(e); if (v > 0 && v <= 65536) ctx->emit_blocks = v; }
    return PM_OK;
}

#define LAUNCH_SCENE_ON(kern, grid, block, ctx, strm, ...)                                                                  \
    do {                                                                                                                    \
        if ((ctx)->scene_mode == SCENE_LDS) hipLaunchKernelGGL(kern<SCENE_LDS>, grid, block, (ctx)->scene_b, strm, (ctx)->gv, __VA_ARGS__); \
        else if ((ctx)->scene_mode == SCENE_GRID) hipLaunchKernelGGL(kern<SCENE_GRID>, grid, block, 0, strm, (ctx)->gv, __VA_ARGS__);       \
        else hipLaunchKernelGGL(kern<SCENE_GLOBAL>, grid, block, 0, strm, (ctx)->gv, __VA_ARGS__);                           \
    } while (0)
#define LAUNCH_SCENE(kern, grid, block, ctx, ...) LAUNCH_SCENE_ON(kern, grid, block, ctx, (ctx)->stream, __VA_ARGS__)

// ------------------------------------------------------------------------------------- emission
// attempts [a0, a0+count) -> first `cap` successes, in attempt order, into d_out, on the slot's stream with the slot's
// scratch; the number of successes in the range lands in *h_got (pinned) when the stream gets there.
// total != NULL: wait for it.
struct EmitScratch { pm_photon *tmp; uint8_t *flag; uint32_t *counts; uint64_t *offs; };   // offs: nb + 1 offsets, then the work counter of k_emit_stream
static int slot_scratch(BuildSlot &sl, uint64_t count, EmitScratch &es) {
    uint32_t nb = (uint32_t)((count + 255) / 256);
    PMCHK(sl.emit_tmp.ensure(count * sizeof(pm_photon)));
    PMCHK(sl.emit_flag.ensure(count));
    PMCHK(sl.emit_counts.ensure((size_t)nb * 4));
    PMCHK(sl.emit_offs.ensure(((size_t)nb + 2) * 8));
    es = {sl.emit_tmp.as<pm_photon>(), sl.emit_flag.as<uint8_t>(), sl.emit_counts.as<uint32_t>(), sl.emit_offs.as<uint64_t>()};
    return PM_OK;
}
constexpr uint64_t EMIT_CHUNK = 1ULL << 24;   // attempts per launch of a map's emission (512 MB of attempt records per build slot)
// d_running != NULL: this range appends behind the ranges emitted before it (k_scan_blocks); *h_got then receives the total so far
static int emit_range_on(pm_ctx *ctx, hipStream_t stream, const EmitScratch &es, int kind, uint64_t seed, const double lc[3], uint64_t a0, uint64_t count,
                         pm_photon *d_out, uint64_t cap, uint64_t *h_got, uint64_t *total, uint64_t *d_running = nullptr, uint32_t *bounds_init = nullptr) {
    uint32_t nb = (uint32_t)((count + 255) / 256);
    uint64_t K = stream_key(seed, kind == PM_MAP_GLOBAL ? KIND_EMIT_GLOBAL : KIND_EMIT_CAUSTIC);
    V3 light = {lc[0], lc[1], lc[2]};
    if (ctx->scene_mode == SCENE_GRID && ctx->emit_stream) {
        // a resident grid of waves that draw attempts from a counter (pm_trace.h)
        uint32_t *next = reinterpret_cast<uint32_t *>(es.offs + nb + 1);
        HIPCHK(hipMemsetAsync(next, 0, 4, stream));
        hipLaunchKernelGGL(k_emit_stream, dim3(std::min<uint32_t>(nb, (uint32_t)ctx->emit_blocks)), dim3(256), 0, stream, ctx->gv, kind, K, light, a0, count, es.tmp, es.flag, next);
        hipLaunchKernelGGL(k_flag_counts, dim3(nb), dim3(256), 0, stream, es.flag, count, es.counts);
    } else if (ctx->emit_run)
        LAUNCH_SCENE_ON(k_emit_run, dim3((uint32_t)((count + 4 * EMIT_RUN - 1) / (4 * EMIT_RUN))), dim3(256), ctx, stream, kind, K, light, a0, count, es.tmp, es.flag, es.counts);
    else
        LAUNCH_SCENE_ON(k_emit, dim3(nb), dim3(256), ctx, stream, kind, K, light, a0, count, es.tmp, es.flag, es.counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, es.counts, es.offs, nb, d_running, bounds_init);
    hipLaunchKernelGGL(k_emit_scatter, dim3(nb), dim3(256), 0, stream, es.tmp, es.flag, es.offs, count, d_out, cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_got, es.offs + nb, 8, hipMemcpyDeviceToHost, stream));
    if (!total) return PM_OK;
    HIPCHK(hipStreamSynchronize(stream));
    *total = *h_got;
    return PM_OK;
}
static int emit_range(pm_ctx *ctx, BuildSlot &sl, int kind, uint64_t seed, const double lc[3], uint64_t a0, uint64_t count, pm_photon *d_out,
                      uint64_t cap, uint64_t *h_got, uint64_t *total, uint64_t *d_running = nullptr, uint32_t *bounds_init = nullptr) {
    if (count == 0) { *h_got = 0; if (total) *total = 0; return PM_OK; }
    if (count > (1ULL << 31)) return fail(PM_EINVAL, "emit_range: at most 2^31 attempts per call");
    EmitScratch es;
    PMCHK(slot_scratch(sl, count, es));
    return emit_range_on(ctx, sl.stream, es, kind, seed, lc, a0, count, d_out, cap, h_got, total, d_running, bounds_init);
}

#ifdef PM_PAIR_STAMPS
extern "C" int pm_pair_ticks(unsigned long long *out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pair_ticks), 8 * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pair_ticks), z, 8 * 8) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef PM_WAVE_TIMES
extern "C" int pm_tile_waves(uint32_t *out, int reset) {     // 2 x 8192 x 4 words per wave, then 2 x 16384 x 2 words per group
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_waves), sizeof(uint32_t) * 2 * 8192 * 4) != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out + 2 * 8192 * 4, HIP_SYMBOL(g_tile_groups), sizeof(uint32_t) * 2 * 16384 * 2) != hipSuccess) return -1;
    if (reset) { static std::vector<uint32_t> z(2 * 8192 * 4, 0u); if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_waves), z.data(), z.size() * 4) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef PM_TRACE_STATS
extern "C" int pm_trace_stats(unsigned long long *out16, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_trace_stats), 16 * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_stats), z, 16 * 8) != hipSuccess) return -1; }
    return 0;
}
#endif
extern "C" int pm_emit_range(pm_ctx *ctx, int32_t map_kind, uint64_t seed, const double light_color[3], uint64_t attempt_begin,
                             uint64_t attempt_count, pm_photon *d_out, uint64_t cap, uint64_t *out_count) {
    if (!ctx || !light_color || !out_count || (!d_out && cap)) return fail(PM_EINVAL, "pm_emit_range: NULL argument");
    if (map_kind != PM_MAP_GLOBAL && map_kind != PM_MAP_CAUSTIC) return fail(PM_EINVAL, "pm_emit_range: bad map_kind");
    if (ctx->nprims == 0) return fail(PM_ESTATE, "pm_emit_range: no scene uploaded");
    PMCHK(set_device(ctx));
    ctx->slot[2].stream = ctx->stream;
    return emit_range(ctx, ctx->slot[2], map_kind, seed, light_color, attempt_begin, attempt_count, d_out, cap, ctx->h_u64, out_count);
}

// pm_emit_range for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc): the same kernels on
// the static scratch array instead of the slot's hipMalloc'ed buffers.  No hipMalloc / hipFree / allocating routine is
// called.  (Not for code under HIP graph capture: the call waits for the stream to learn the count.)  Every call on a GPU shares that GPU's array: a call overwrites all of the scratch it reads (flags, block counts,
// offsets, records of its own attempts), so nothing has to be cleared, and two calls that overlap on one GPU race.
extern "C" int pm_emit_range_static(pm_ctx *ctx, int32_t map_kind, uint64_t seed, const double light_color[3], uint64_t attempt_begin,
                                    uint64_t attempt_count, pm_photon *d_out, uint64_t cap, uint64_t *out_count) {
    if (!ctx || !light_color || !out_count || (!d_out && cap)) return fail(PM_EINVAL, "pm_emit_range_static: NULL argument");
    if (map_kind != PM_MAP_GLOBAL && map_kind != PM_MAP_CAUSTIC) return fail(PM_EINVAL, "pm_emit_range_static: bad map_kind");
    if (attempt_count > ES_MAX)
        return fail(PM_EINVAL, "pm_emit_range_static: at most " + std::to_string(ES_MAX) + " attempts per call (PM_EMIT_STATIC_MAX_ATTEMPTS: the static scratch "
                               "array holds no more); split the range, or use pm_emit_range");
    if (ctx->nprims == 0) return fail(PM_ESTATE, "pm_emit_range_static: no scene uploaded");
    if (!ctx->static_scratch) return fail(PM_ESTATE, "pm_emit_range_static: the context has no address for the static scratch array");
    PMCHK(set_device(ctx));
    if (attempt_count == 0) { *out_count = 0; return PM_OK; }
    unsigned char *sp = ctx->static_scratch;   // the device address, never the array's name
    const EmitScratch es = {reinterpret_cast<pm_photon *>(sp + ES_OFF_TMP), sp + ES_OFF_FLAG, reinterpret_cast<uint32_t *>(sp + ES_OFF_COUNTS),
                            reinterpret_cast<uint64_t *>(sp + ES_OFF_OFFS)};
    return emit_range_on(ctx, ctx->stream, es, map_kind, seed, light_color, attempt_begin, attempt_count, d_out, cap, ctx->h_u64, out_count);
}

extern "C" int pm_emit_shard(pm_ctx *ctx, int32_t map_kind, uint64_t seed, const double light_color[3], uint64_t attempt_begin,
                             uint64_t attempt_count, void *d_shard, uint64_t cap) {
    if (!ctx || !light_color || !d_shard) return fail(PM_EINVAL, "pm_emit_shard: NULL argument");
    if (map_kind != PM_MAP_GLOBAL && map_kind != PM_MAP_CAUSTIC) return fail(PM_EINVAL, "pm_emit_shard: bad map_kind");
    if (ctx->nprims == 0) return fail(PM_ESTATE, "pm_emit_shard: no scene uploaded");
    PMCHK(set_device(ctx));
    BuildSlot &sl = ctx->slot[2];
    sl.stream = ctx->stream;
    const uint64_t head[4] = {0, cap, 0, 0};
    HIPCHK(hipMemcpyAsync(d_shard, head, sizeof(head), hipMemcpyHostToDevice, ctx->stream));
    if (attempt_count == 0) return PM_OK;
    PMCHK(emit_range(ctx, sl, map_kind, seed, light_color, attempt_begin, attempt_count, reinterpret_cast<pm_photon *>((char *)d_shard + 32), cap, ctx->h_u64, nullptr));
    // the count of successes (it may exceed cap: pm_map_build_shards reports that) goes into the header, on the device
    uint32_t nb = (uint32_t)((attempt_count + 255) / 256);
    HIPCHK(hipMemcpyAsync(d_shard, sl.emit_offs.as<uint64_t>() + nb, 8, hipMemcpyDeviceToDevice, ctx->stream));
    return PM_OK;
}

// ---------------------------------------------------------------------------------------- build
static int build_index(pm_map *map, BuildSlot &sl, bool bounds_reset = false) {
    long long n = map->n;
    MapView &v = map->view;
    memset(&v, 0, sizeof(v));
    v.n = n;
    if (n == 0) return PM_OK;
    if (n > (1LL << 31) - 64) return fail(PM_EINVAL, "pm_map_build: at most 2^31-64 photons per map");
    const pm_photon *rec = map->records.as<pm_photon>();
    PMCHK(sl.keys_in.ensure((size_t)n * 8));
    PMCHK(sl.vals_in.ensure((size_t)n * 4)); PMCHK(sl.vals_out.ensure((size_t)n * 4));
    PMCHK(sl.bounds.ensure(6 * 4));
    PMCHK(map->pos4.ensure((size_t)n * 16)); PMCHK(map->rest.ensure((size_t)n * 16));
    // (the six bounds words were reset by the emission's scan kernel when this map was emitted here; otherwise now)
    if (!bounds_reset) {
        HIPCHK(hipMemsetAsync(sl.bounds.p, 0xFF, 12, sl.stream));
        HIPCHK(hipMemsetAsync(sl.bounds.as<unsigned char>() + 12, 0, 12, sl.stream));
    }
    unsigned nbk = (unsigned)std::min<long long>((n + 255) / 256, 512);
    unsigned nb = (unsigned)((n + 255) / 256);
    const uint32_t sort_blocks = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
    PMCHK(sl.sort_tmp.ensure(sort_scratch_words((uint32_t)n) * 4));
    PMCHK(map->keys.ensure((size_t)n * 8));
    PMCHK(map->grid.ensure(sizeof(GridParams) + 16));
    // linear octree: the finest table level holds about two cells per photon
    int lmax = 0;
    while (lmax < 8 && (1LL << (3 * (lmax + 1))) <= 4 * n) lmax++;
    const long long n_entries = (1LL << (3 * lmax)) + 1;
    const uint32_t fill_blocks = (uint32_t)((n_entries + FILL_TILE - 1) / FILL_TILE);
    PMCHK(map->cells.ensure((size_t)n_entries * 4 + (size_t)fill_blocks * 4));
    PMCHK(map->cbox.ensure((size_t)box_off(lmax + 1) * 8));
    uint32_t *cells = map->cells.as<uint32_t>(), *bmin = cells + n_entries;
    hipLaunchKernelGGL(k_bounds, dim3(nbk), dim3(256), 0, sl.stream, rec, n, sl.bounds.as<uint32_t>());
    // keys + the sort's first histogram + the grid numbers; 30-bit radix sort (pm_sort.h) straight into the map's key array
    hipLaunchKernelGGL(k_morton, dim3(sort_blocks), dim3(256), 0, sl.stream, rec, n, sl.bounds.as<uint32_t>(), sl.keys_in.as<uint64_t>(), sl.vals_in.as<uint32_t>(),
                       sort_blocks, sl.sort_tmp.as<uint32_t>(), map->grid.as<GridParams>());
    HIPCHK(radix_sort_pairs_30(sl.stream, sl.keys_in.as<uint64_t>(), sl.vals_in.as<uint32_t>(), map->keys.as<uint64_t>(), sl.vals_out.as<uint32_t>(),
                               sl.sort_tmp.as<uint32_t>(), (uint32_t)n, nullptr, true));
    // the cell table: preset, first photons of the occupied cells (k_permute), empty cells filled from the right
    HIPCHK(hipMemsetAsync(cells, 0xFF, (size_t)n_entries * 4, sl.stream));
    hipLaunchKernelGGL(k_permute, dim3(nb), dim3(256), 0, sl.stream, rec, sl.vals_out.as<uint32_t>(), map->keys.as<uint64_t>(), n, lmax, map->pos4.as<float4>(),
                       map->rest.as<uint4>(), cells);
    hipLaunchKernelGGL(k_fill_mins, dim3(fill_blocks), dim3(256), 0, sl.stream, cells, n_entries, bmin);
    hipLaunchKernelGGL(k_fill_cells, dim3(fill_blocks), dim3(256), 0, sl.stream, cells, n_entries, bmin, fill_blocks);
    // cell boxes: the finest level and up to three levels above it per subtree, then the few levels on top
    const int depth = std::min(3, lmax);
    hipLaunchKernelGGL(k_cell_boxes_tree, dim3((unsigned)(1LL << (3 * (lmax - depth)))), dim3(256), 0, sl.stream, map->pos4.as<float4>(), cells, lmax, depth,
                       sl.bounds.as<uint32_t>(), map->cbox.as<uint64_t>());
    if (lmax - depth - 1 >= 0) hipLaunchKernelGGL(k_cell_boxes_top, dim3(1), dim3(256), 0, sl.stream, lmax - depth - 1, map->cbox.as<uint64_t>());
    v.grid = map->grid.as<GridParams>();   // (glo / gext / gu stay zero on the host: kernels load them from here)
    v.lmax = lmax; v.cells = map->cells.as<uint32_t>(); v.keys = map->keys.as<uint64_t>(); v.cbox = map->cbox.as<uint64_t>();
    v.pos4 = map->pos4.as<float4>(); v.rest = map->rest.as<uint4>();
    HIPCHK(hipGetLastError());
    return PM_OK;   // nothing here waits for the device: the caller synchronises once
}

// a map object: a recycled one (device buffers of a destroyed map of this context) when there is one
static pm_map *new_map(pm_ctx *ctx) {
    pm_map *m = nullptr;
    if (!ctx->spare.empty()) { m = ctx->spare.back(); ctx->spare.pop_back(); }
    else m = new (std::nothrow) pm_map();
    if (!m) return nullptr;
    if (!m->ready && (hipEventCreate(&m->ready) != hipSuccess || hipEventCreate(&m->t_begin) != hipSuccess || hipEventCreate(&m->t_emitted) != hipSuccess)) { free_map(m); return nullptr; }
    if (!m->h_got && hipHostMalloc((void **)&m->h_got, 2 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { free_map(m); return nullptr; }
    m->ctx = ctx; m->pending = false; m->failed = PM_OK; m->fail_msg.clear();
    return m;
}
// the next build slot (0 / 1 in turn), its stream ordered behind what the caller's stream holds now
static int stamp(hipEvent_t a, hipEvent_t b, hipStream_t st) {
    HIPCHK(hipEventRecord(a, st));
    if (b) HIPCHK(hipEventRecord(b, st));
    return PM_OK;
}
static int take_slot(pm_ctx *ctx, int &slot) {
    slot = ctx->next_slot;
    ctx->next_slot ^= 1;
    HIPCHK(hipEventRecord(ctx->ev_order, ctx->stream));
    HIPCHK(hipStreamWaitEvent(ctx->slot[slot].stream, ctx->ev_order, 0));
    return PM_OK;
}

// Wait for a map whose emission and build were left on a slot's stream and make sure it holds its n photons.
static int map_top_up(pm_map *map, uint64_t have) {
    pm_ctx *ctx = map->ctx;
    const uint64_t n = (uint64_t)map->n;
    const double yield = ctx->yield[map->kind];
    BuildSlot &sl = ctx->slot[2];
    sl.stream = ctx->stream;
    int empty_batches = 0;
    while (have < n) {
        uint64_t want = n - have;
        uint64_t batch = std::min<uint64_t>((uint64_t)((double)want / yield) + 4096, EMIT_CHUNK);
        uint64_t got = 0;
        PMCHK(emit_range(ctx, sl, map->kind, map->seed, map->lc, map->next_attempt, batch, map->records.as<pm_photon>() + have, want, ctx->h_u64, &got));
        map->next_attempt += batch;
        have += std::min<uint64_t>(got, want);
        if (got == 0 && ++empty_batches > 8) return fail(PM_ESTATE, "pm_map_create: no emission attempt stores a photon in this scene");
    }
    PMCHK(build_index(map, sl));
    HIPCHK(hipStreamSynchronize(sl.stream));
    return PM_OK;
}
static int map_failed(pm_map *map, int code, const std::string &msg) { map->failed = code; map->fail_msg = msg; return fail(code, msg); }
static int map_ready(pm_map *map, bool *rebuilt = nullptr) {
    if (!map) return PM_OK;
    if (map->failed != PM_OK) return fail(map->failed, map->fail_msg);
    if (!map->pending) return PM_OK;
    pm_ctx *ctx = map->ctx;
    map->pending = false;
    if (hipEventSynchronize(map->ready) != hipSuccess) return map_failed(map, PM_EHIP, "photon map: emission or build failed on the device");
    if (map->kind == -2) {             // assembled from shards: they must have held n photons, none cut off
        if (map->h_got[1]) return map_failed(map, PM_ESTATE, "pm_map_build_shards: a shard overflowed its capacity");
        if (*map->h_got < (uint64_t)map->n) return map_failed(map, PM_ESTATE, "pm_map_build_shards: the shards hold fewer than n photons");
        return PM_OK;
    }
    if (map->kind < 0) return PM_OK;   // built from records: nothing to check
    const uint64_t n = (uint64_t)map->n;
    uint64_t have = std::min<uint64_t>(*map->h_got, n);
    if (map->next_attempt >= 4096) {
        // what this scene really yields, less four standard deviations and a hundredth: the next map's range is sized by it
        const double a = (double)map->next_attempt, p = (double)*map->h_got / a;
        ctx->yield[map->kind] = std::max(1e-4, p - 4.0 * sqrt(p * (1.0 - p) / a) - 0.01 * p);
        ctx->yield_known[map->kind] = true;
    }
    if (have >= n) return PM_OK;
    // the ranges fell short of n photons (a scene that stores far fewer photons per attempt than the reference's):
    // keep emitting, synchronously on the caller's stream, then build again
    int rc = map_top_up(map, have);
    if (rc != PM_OK) { map->failed = rc; map->fail_msg = g_err; }
    else if (rebuilt) *rebuilt = true;
    return rc;
}
extern "C" int pm_map_build(pm_ctx *ctx, const pm_photon *records, int64_t n, int32_t records_on_device, pm_map **out) {
    if (!ctx || !out || n < 0 || (n > 0 && !records)) return fail(PM_EINVAL, "pm_map_build: bad argument");
    PMCHK(set_device(ctx));
    pm_map *map = new_map(ctx);
    if (!map) return fail(PM_ENOMEM, "out of host memory");
    map->n = n; map->kind = -1;
    int rc = map->records.ensure((size_t)std::max<int64_t>(n, 1) * sizeof(pm_photon));
    if (rc == PM_OK && n > 0) {
        // the copy is ordered on the caller's stream (the caller may reuse `records` behind this call); the build follows
        // on a slot's stream and is waited for by the first use of the map
        hipError_t e = hipMemcpyAsync(map->records.p, records, (size_t)n * sizeof(pm_photon),
                                      records_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = fail(PM_EHIP, std::string("copy of photon records failed: ") + hipGetErrorString(e));
    }
    if (rc == PM_OK) rc = take_slot(ctx, map->slot);
    if (rc == PM_OK) rc = stamp(map->t_begin, map->t_emitted, ctx->slot[map->slot].stream);
    if (rc == PM_OK) rc = build_index(map, ctx->slot[map->slot]);
    if (rc == PM_OK && hipEventRecord(map->ready, ctx->slot[map->slot].stream) != hipSuccess) rc = fail(PM_EHIP, "pm_map_build: hipEventRecord failed");
    if (rc != PM_OK) { pm_map_destroy(map); return rc; }
    map->pending = true;
    *out = map;
    return PM_OK;
}

extern "C" int pm_map_build_shards(pm_ctx *ctx, const void *d_shards, int32_t world, uint64_t shard_stride, int64_t n, pm_map **out) {
    if (!ctx || !out || n < 0 || !d_shards || world < 1 || world > 64 || shard_stride < 32 || (shard_stride & 15))
        return fail(PM_EINVAL, "pm_map_build_shards: bad argument (1..64 shards, stride a multiple of 16 bytes)");
    PMCHK(set_device(ctx));
    pm_map *map = new_map(ctx);
    if (!map) return fail(PM_ENOMEM, "out of host memory");
    map->n = n; map->kind = -2;
    int rc = map->records.ensure((size_t)std::max<int64_t>(n, 1) * sizeof(pm_photon));
    if (rc == PM_OK) rc = map->grid.ensure(sizeof(GridParams) + 16);
    if (rc == PM_OK && n > 0) {
        // assembled on the caller's stream (the caller may reuse the shards behind this call); the count check travels
        // with the map and is made by its first use
        unsigned long long *status = reinterpret_cast<unsigned long long *>(map->grid.as<char>() + sizeof(GridParams));
        hipLaunchKernelGGL(k_assemble_shards, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned char *>(d_shards),
                           (int)world, (unsigned long long)shard_stride, (long long)n, map->records.as<pm_photon>(), status);
        if (hipMemcpyAsync(map->h_got, status, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(PM_EHIP, "pm_map_build_shards: copy failed");
    } else if (rc == PM_OK) { map->h_got[0] = 0; map->h_got[1] = 0; }
    if (rc == PM_OK) rc = take_slot(ctx, map->slot);
    if (rc == PM_OK) rc = stamp(map->t_begin, map->t_emitted, ctx->slot[map->slot].stream);
    if (rc == PM_OK) rc = build_index(map, ctx->slot[map->slot]);
    if (rc == PM_OK && hipEventRecord(map->ready, ctx->slot[map->slot].stream) != hipSuccess) rc = fail(PM_EHIP, "pm_map_build_shards: hipEventRecord failed");
    if (rc != PM_OK) { pm_map_destroy(map); return rc; }
    map->pending = true;
    *out = map;
    return PM_OK;
}

extern "C" int pm_map_create(pm_ctx *ctx, int32_t map_kind, int64_t n, uint64_t seed, const double light_color[3], pm_map **out,
                             uint64_t *attempts_used) {
    if (!ctx || !out || n < 0 || !light_color) return fail(PM_EINVAL, "pm_map_create: bad argument");
    if (map_kind != PM_MAP_GLOBAL && map_kind != PM_MAP_CAUSTIC) return fail(PM_EINVAL, "pm_map_create: bad map_kind");
    if (ctx->nprims == 0) return fail(PM_ESTATE, "pm_map_create: no scene uploaded");
    PMCHK(set_device(ctx));
    pm_map *map = new_map(ctx);
    if (!map) return fail(PM_ENOMEM, "out of host memory");
    map->n = n; map->kind = map_kind; map->seed = seed; map->next_attempt = 0;
    for (int a = 0; a < 3; a++) map->lc[a] = light_color[a];
    int rc = map->records.ensure((size_t)std::max<int64_t>(n, 1) * sizeof(pm_photon));
    // Consecutive attempt ranges; the result (first n successes in attempt order) does not depend on how they are cut.
    // The first range is sized to hold n successes with room to spare, and the index is built right behind it on a
    // slot's stream without waiting for the count.  With attempts_used == NULL the call returns at once: the first
    // use of the map waits (map_ready), so two maps are emitted and built side by side, beside the first passes of the
    // frame that needs them.  When the guess does not hold (it does for any scene close to the reference's) map_ready
    // emits more and builds again.
    const double yield = ctx->yield[map_kind];
    if (rc == PM_OK) rc = take_slot(ctx, map->slot);
    if (rc == PM_OK) {
        BuildSlot &sl = ctx->slot[map->slot];
        // as many attempts as n photons are expected to take, in launches of at most EMIT_CHUNK chained on the device (a 10M-photon
        // caustic map is 28M attempts): no count reaches the host before the index is built
        const uint64_t total = n > 0 ? std::min<uint64_t>((uint64_t)((double)n / yield) + 4096, 1ULL << 36) : 0;
        map->h_got[0] = 0;
        rc = stamp(map->t_begin, nullptr, sl.stream);
        if (rc == PM_OK) rc = sl.bounds.ensure(6 * 4);
        if (total > EMIT_CHUNK) {
            rc = sl.emit_run.ensure(8);
            if (rc == PM_OK && hipMemsetAsync(sl.emit_run.p, 0, 8, sl.stream) != hipSuccess) rc = fail(PM_EHIP, "pm_map_create: hipMemsetAsync failed");
        }
        for (uint64_t a0 = 0; rc == PM_OK && a0 < total; a0 += EMIT_CHUNK)
            rc = emit_range(ctx, sl, map_kind, seed, light_color, a0, std::min<uint64_t>(EMIT_CHUNK, total - a0), map->records.as<pm_photon>(), (uint64_t)n, map->h_got, nullptr,
                            total > EMIT_CHUNK ? sl.emit_run.as<uint64_t>() : nullptr, sl.bounds.as<uint32_t>());
        map->next_attempt = total;
        if (rc == PM_OK) rc = stamp(map->t_emitted, nullptr, sl.stream);
        if (rc == PM_OK) rc = build_index(map, sl, total > 0);
        if (rc == PM_OK && hipEventRecord(map->ready, sl.stream) != hipSuccess) rc = fail(PM_EHIP, "pm_map_create: hipEventRecord failed");
    }
    if (rc == PM_OK) { map->pending = true; if (attempts_used) rc = map_ready(map); }
    if (rc != PM_OK) { pm_map_destroy(map); return rc; }
    if (attempts_used) *attempts_used = map->next_attempt;   // upper bound: attempts issued (whole ranges)
    *out = map;
    return PM_OK;
}

extern "C" int pm_map_size(const pm_map *map, int64_t *n) {
    if (!map || !n) return fail(PM_EINVAL, "pm_map_size: NULL argument");
    *n = map->n;
    return PM_OK;
}
extern "C" int pm_map_times(pm_map *map, float *ms_emit, float *ms_build) {
    if (!map || !ms_emit || !ms_build) return fail(PM_EINVAL, "pm_map_times: NULL argument");
    PMCHK(set_device(map->ctx));
    PMCHK(map_ready(map));
    HIPCHK(hipEventSynchronize(map->ready));
    *ms_emit = *ms_build = 0.f;
    if (hipEventElapsedTime(ms_emit, map->t_begin, map->t_emitted) != hipSuccess) *ms_emit = 0.f;
    if (hipEventElapsedTime(ms_build, map->t_emitted, map->ready) != hipSuccess) *ms_build = 0.f;
    return PM_OK;
}
extern "C" int pm_map_download(pm_map *map, pm_photon *host_out) {
    if (!map || (!host_out && map->n)) return fail(PM_EINVAL, "pm_map_download: NULL argument");
    PMCHK(set_device(map->ctx));
    PMCHK(map_ready(map));
    if (map->n) {
        HIPCHK(hipMemcpyAsync(host_out, map->records.p, (size_t)map->n * sizeof(pm_photon), hipMemcpyDeviceToHost, map->ctx->stream));
        HIPCHK(hipStreamSynchronize(map->ctx->stream));
    }
    return PM_OK;
}
static void free_map(pm_map *map) {
    map->records.release(); map->pos4.release(); map->rest.release(); map->cells.release(); map->keys.release(); map->cbox.release(); map->grid.release();
    if (map->ready) (void)hipEventDestroy(map->ready);
    if (map->t_begin) (void)hipEventDestroy(map->t_begin);
    if (map->t_emitted) (void)hipEventDestroy(map->t_emitted);
    if (map->h_got) (void)hipHostFree(map->h_got);
    delete map;
}
extern "C" int pm_map_destroy(pm_map *map) {
    if (!map) return PM_OK;
    pm_ctx *ctx = map->ctx;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        if (map->pending) { (void)hipEventSynchronize(map->ready); map->pending = false; }   // its build may still be running
        // and so may the gathers of the latest frame that read it (a frame that ended in an error has already waited for
        // them; this is for callers that destroy a map behind a frame they have not waited for): the end of that frame, no
        // more -- the other map's build or a later frame may be queued on the same streams and are none of this map's business
        if (map->used_in_frame > ctx->frame_waited) {
            if (map->used_in_frame == ctx->frame_seq) { (void)hipEventSynchronize(ctx->ev_frame); ctx->frame_waited = ctx->frame_seq; }
            else { (void)hipStreamSynchronize(ctx->stream); ctx->frame_waited = ctx->frame_seq; }   // (an older frame: its event has been recorded over)
        }
    }
    // keep the buffers of up to four maps for the next ones (a frame rebuilds both maps; hipFree / hipMalloc of 100 MB cost more than the build)
    if (ctx && ctx->spare.size() < 4) { map->n = 0; memset(&map->view, 0, sizeof(map->view)); ctx->spare.push_back(map); }
    else free_map(map);
    return PM_OK;
}

// --------------------------------------------------------------------------------------- gather
// grid for the wave-per-query kernels: a multiple of 8 blocks (one group per XCD), at most 8 blocks per CU
static void gather_grid(long long count, unsigned &blocks, uint32_t &waves_total) {
    long long want = (count + 3) / 4;
    blocks = (unsigned)std::max<long long>(8, std::min<long long>((want + 7) / 8 * 8, 256 * 8));
    waves_total = blocks * 4;
}

static int knn_query_impl(pm_ctx *ctx, pm_map *map, int64_t m, const double *points, int32_t *idx, double *d2, int32_t *found, uint32_t *work) {
    if (!ctx || !map || m < 0 || (m && (!points || !idx || !d2 || !found))) return fail(PM_EINVAL, "pm_knn_query: bad argument");
    if (m == 0) return PM_OK;
    PMCHK(set_device(ctx));
    PMCHK(map_ready(map));
    DevBuf dp, di, dd, df, dw;
    int rc = dp.ensure((size_t)m * 24);
    if (rc == PM_OK) rc = di.ensure((size_t)m * KNN_K * 4);
    if (rc == PM_OK) rc = dd.ensure((size_t)m * KNN_K * 8);
    if (rc == PM_OK) rc = df.ensure((size_t)m * 4);
    if (rc == PM_OK && work) rc = dw.ensure((size_t)m * 16);
    hipError_t e = hipSuccess;
    if (rc == PM_OK) {
        e = hipMemcpyAsync(dp.p, points, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream);
        unsigned blocks; uint32_t waves;
        gather_grid(m, blocks, waves);
        if (e == hipSuccess) hipLaunchKernelGGL(k_knn_query, dim3(blocks), dim3(256), 0, ctx->stream, map->view, dp.as<double>(), (long long)m,
                                                di.as<int32_t>(), dd.as<double>(), df.as<int32_t>(), work ? dw.as<uint32_t>() : nullptr);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(idx, di.p, (size_t)m * KNN_K * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d2, dd.p, (size_t)m * KNN_K * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(found, df.p, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && work) e = hipMemcpyAsync(work, dw.p, (size_t)m * 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    dp.release(); di.release(); dd.release(); df.release(); dw.release();
    if (rc != PM_OK) return rc;
    if (e != hipSuccess) return fail(PM_EHIP, std::string("pm_knn_query: ") + hipGetErrorString(e));
    return PM_OK;
}

extern "C" int pm_knn_query(pm_ctx *ctx, pm_map *map, int64_t m, const double *points, int32_t *idx, double *d2, int32_t *found) {
    return knn_query_impl(ctx, map, m, points, idx, d2, found, nullptr);
}
extern "C" int pm_knn_profile(pm_ctx *ctx, pm_map *map, int64_t m, const double *points, int32_t *idx, double *d2, int32_t *found, uint32_t *work) {
    if (m && !work) return fail(PM_EINVAL, "pm_knn_profile: work is NULL");
    return knn_query_impl(ctx, map, m, points, idx, d2, found, work);
}

extern "C" int pm_knn_gather(pm_ctx *ctx, pm_map *map, int32_t map_kind, int64_t m, const double *points, const double *normals,
                             double *radiance, int32_t on_device) {
    if (!ctx || !map || m < 0 || (m && (!points || !normals || !radiance))) return fail(PM_EINVAL, "pm_knn_gather: bad argument");
    if (map_kind != PM_MAP_GLOBAL && map_kind != PM_MAP_CAUSTIC) return fail(PM_EINVAL, "pm_knn_gather: bad map_kind");
    if (m == 0) return PM_OK;
    PMCHK(set_device(ctx));
    PMCHK(map_ready(map));
    DevBuf dp, dn, dr;
    const double *p = points, *nn = normals;
    double *r = radiance;
    int rc = PM_OK;
    hipError_t e = hipSuccess;
    if (!on_device) {
        rc = dp.ensure((size_t)m * 24);
        if (rc == PM_OK) rc = dn.ensure((size_t)m * 24);
        if (rc == PM_OK) rc = dr.ensure((size_t)m * 24);
        if (rc == PM_OK) {
            e = hipMemcpyAsync(dp.p, points, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dn.p, normals, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream);
            p = dp.as<double>(); nn = dn.as<double>(); r = dr.as<double>();
        }
    }
    if (rc == PM_OK && e == hipSuccess) {
        unsigned blocks; uint32_t waves;
        gather_grid(m, blocks, waves);
        if (map_kind == PM_MAP_GLOBAL) hipLaunchKernelGGL(k_gather_points<0>, dim3(blocks), dim3(256), 0, ctx->stream, map->view, p, nn, r, (long long)m);
        else hipLaunchKernelGGL(k_gather_points<1>, dim3(blocks), dim3(256), 0, ctx->stream, map->view, p, nn, r, (long long)m);
        e = hipGetLastError();
        if (e == hipSuccess && !on_device) e = hipMemcpyAsync(radiance, dr.p, (size_t)m * 24, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    dp.release(); dn.release(); dr.release();
    if (rc != PM_OK) return rc;
    if (e != hipSuccess) return fail(PM_EHIP, std::string("pm_knn_gather: ") + hipGetErrorString(e));
    return PM_OK;
}

// --------------------------------------------------------------------------------------- frames
static int make_frame_params(pm_ctx *ctx, const pm_render_params *rp, FrameParams &fp) {
    if (!ctx || !rp) return fail(PM_EINVAL, "NULL argument");
    if (ctx->nprims == 0) return fail(PM_ESTATE, "no scene uploaded");
    if (rp->width <= 0 || rp->height <= 0 || (long long)rp->width * rp->height > (1LL << 27))
        return fail(PM_EINVAL, "pane must be between 1x1 and 2^27 pixels (pixel index * 32 is the RNG item)");
    if (rp->world < 1 || rp->rank < 0 || rp->rank >= rp->world) return fail(PM_EINVAL, "bad rank/world");
    fp.width = rp->width; fp.height = rp->height;
    fp.cam = {rp->cam[0], rp->cam[1], rp->cam[2]};
    fp.light = {rp->light_color[0], rp->light_color[1], rp->light_color[2]};
    fp.K = stream_key(rp->seed, KIND_PIXEL);
    fp.flags = rp->flags; fp.rank = rp->rank; fp.world = rp->world;
    return PM_OK;
}

extern "C" int pm_primary_hits(pm_ctx *ctx, const pm_render_params *rp, int32_t y0, int32_t y1, int32_t *ids, double *points, double *normals) {
    FrameParams fp;
    PMCHK(make_frame_params(ctx, rp, fp));
    if (!ids || y0 < 0 || y1 > rp->height || y0 > y1) return fail(PM_EINVAL, "pm_primary_hits: bad row range or ids is NULL");
    long long total = (long long)(y1 - y0) * rp->width;
    if (total == 0) return PM_OK;
    PMCHK(set_device(ctx));
    DevBuf di, dp, dn;
    int rc = di.ensure((size_t)total * 4);
    if (rc == PM_OK && points) rc = dp.ensure((size_t)total * 24);
    if (rc == PM_OK && normals) rc = dn.ensure((size_t)total * 24);
    hipError_t e = hipSuccess;
    if (rc == PM_OK) {
        LAUNCH_SCENE(k_primary_hits, dim3((unsigned)((total + 255) / 256)), dim3(256), ctx, fp, (int)y0, (int)y1, di.as<int32_t>(),
                     points ? dp.as<double>() : nullptr, normals ? dn.as<double>() : nullptr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(ids, di.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && points) e = hipMemcpyAsync(points, dp.p, (size_t)total * 24, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && normals) e = hipMemcpyAsync(normals, dn.p, (size_t)total * 24, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    di.release(); dp.release(); dn.release();
    if (rc != PM_OK) return rc;
    if (e != hipSuccess) return fail(PM_EHIP, std::string("pm_primary_hits: ") + hipGetErrorString(e));
    return PM_OK;
}

extern "C" int pm_intersect_rays(pm_ctx *ctx, int64_t n, const double *src, const double *dir, const double *dt, const int32_t *prim, const int32_t *exclude,
                                 int32_t *hit, double *res, double *nrm) {
    if (!ctx || n < 0 || (n && (!src || !dir || !hit || !res || !nrm))) return fail(PM_EINVAL, "pm_intersect_rays: bad argument");
    if (ctx->nprims == 0) return fail(PM_ESTATE, "pm_intersect_rays: no scene uploaded");
    if (n == 0) return PM_OK;
    if (n > (1LL << 31)) return fail(PM_EINVAL, "pm_intersect_rays: at most 2^31 rays per call");
    // a primitive index beyond the scene would read past the scene block on the device
    if (prim) for (int64_t i = 0; i < n; i++) if (prim[i] >= ctx->nprims) return fail(PM_EINVAL, "pm_intersect_rays: prim[" + std::to_string(i) + "] is not a primitive of the scene");
    PMCHK(set_device(ctx));
    DevBuf ds, dd, dtb, dp, de, dh, dr, dn;
    int rc = ds.ensure((size_t)n * 24);
    if (rc == PM_OK) rc = dd.ensure((size_t)n * 24);
    if (rc == PM_OK && dt) rc = dtb.ensure((size_t)n * 8);
    if (rc == PM_OK && prim) rc = dp.ensure((size_t)n * 4);
    if (rc == PM_OK && exclude) rc = de.ensure((size_t)n * 4);
    if (rc == PM_OK) rc = dh.ensure((size_t)n * 4);
    if (rc == PM_OK) rc = dr.ensure((size_t)n * 24);
    if (rc == PM_OK) rc = dn.ensure((size_t)n * 24);
    hipError_t e = hipSuccess;
    if (rc == PM_OK) {
        hipStream_t st = ctx->stream;
        e = hipMemcpyAsync(ds.p, src, (size_t)n * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dd.p, dir, (size_t)n * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && dt) e = hipMemcpyAsync(dtb.p, dt, (size_t)n * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && prim) e = hipMemcpyAsync(dp.p, prim, (size_t)n * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && exclude) e = hipMemcpyAsync(de.p, exclude, (size_t)n * 4, hipMemcpyHostToDevice, st);
        // the caller's res / nrm travel to the device first: a miss leaves them as they were (the reference's contract)
        if (e == hipSuccess) e = hipMemcpyAsync(dr.p, res, (size_t)n * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dn.p, nrm, (size_t)n * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            LAUNCH_SCENE(k_intersect_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), ctx, (long long)n, ds.as<double>(), dd.as<double>(), dt ? dtb.as<double>() : nullptr,
                         prim ? dp.as<int32_t>() : nullptr, exclude ? de.as<int32_t>() : nullptr, dh.as<int32_t>(), dr.as<double>(), dn.as<double>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hit, dh.p, (size_t)n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(res, dr.p, (size_t)n * 24, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(nrm, dn.p, (size_t)n * 24, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    DevBuf *all[] = {&ds, &dd, &dtb, &dp, &de, &dh, &dr, &dn};
    for (DevBuf *b : all) b->release();
    if (rc != PM_OK) return rc;
    if (e != hipSuccess) return fail(PM_EHIP, std::string("pm_intersect_rays: ") + hipGetErrorString(e));
    return PM_OK;
}

extern "C" int pm_downsample(pm_ctx *ctx, const uint32_t *pane, int32_t pane_on_device, int32_t width, int32_t height, int32_t sample_ratio, int32_t flip,
                             uint32_t *out, int32_t out_on_device) {
    if (!ctx || !pane || !out) return fail(PM_EINVAL, "pm_downsample: NULL argument");
    if (width <= 0 || height <= 0 || sample_ratio < 1 || sample_ratio > 4096 || (long long)width * sample_ratio * (long long)height * sample_ratio > (1LL << 31))
        return fail(PM_EINVAL, "pm_downsample: need width, height > 0, 1 <= sample_ratio <= 4096 and a pane of at most 2^31 pixels");
    PMCHK(set_device(ctx));
    const size_t in_b = (size_t)width * sample_ratio * height * sample_ratio * 4, out_b = (size_t)width * height * 4;
    const uint32_t *d_in = pane;
    uint32_t *d_out = out;
    if (!pane_on_device) {
        PMCHK(ctx->pane.ensure(in_b));
        HIPCHK(hipMemcpyAsync(ctx->pane.p, pane, in_b, hipMemcpyHostToDevice, ctx->stream));
        d_in = ctx->pane.as<uint32_t>();
    }
    if (!out_on_device) { PMCHK(ctx->sample.ensure(out_b)); d_out = ctx->sample.as<uint32_t>(); }
    hipLaunchKernelGGL(k_downsample, dim3((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4)), dim3(256), 0, ctx->stream, d_in, (int)width, (int)height,
                       (int)sample_ratio, (int)flip, d_out);
    HIPCHK(hipGetLastError());
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

static int ensure_sp(pm_ctx *ctx, uint32_t cap) {
    if (cap <= ctx->sp_cap) return PM_OK;
    size_t c = (size_t)cap + cap / 8 + 1024;
    PMCHK(ctx->geo.ensure(c * 6 * 8)); PMCHK(ctx->obj.ensure(c * 4)); PMCHK(ctx->pixel.ensure(c * 4));
    PMCHK(ctx->node.ensure(c * 4)); PMCHK(ctx->tabslot.ensure(c * 4)); PMCHK(ctx->direct.ensure(c * 8));
    PMCHK(ctx->radg.ensure(c * 3 * 8)); PMCHK(ctx->radc.ensure(c * 3 * 8)); PMCHK(ctx->color.ensure(c * 4));
    ctx->sp_cap = (uint32_t)c;
    return PM_OK;
}
static SPQueue sp_queue(pm_ctx *ctx) {
    SPQueue q;
    q.geo = ctx->geo.as<double>(); q.obj = ctx->obj.as<int32_t>(); q.pixel = ctx->pixel.as<uint32_t>();
    q.node = ctx->node.as<uint32_t>(); q.tabslot = ctx->tabslot.as<uint32_t>(); q.direct = ctx->direct.as<double>();
    q.rad_g = ctx->radg.as<double>(); q.rad_c = ctx->radc.as<double>(); q.color = ctx->color.as<uint32_t>();
    q.overflow = &ctx->ctr.as<Counters>()->overflow;
    q.cap = ctx->sp_cap;
    // option sp_cap_limit (tests): pretend the queue is this small, to show that a frame that overflows it is reported, not delivered
    if (ctx->sp_cap_limit > 0 && (uint32_t)ctx->sp_cap_limit < q.cap) q.cap = (uint32_t)ctx->sp_cap_limit;
    return q;
}

// resident blocks of the gather kernels (they are persistent: work comes from device counters), a multiple of 8 so that
// every XCD gets the same number
static int gather_blocks(pm_ctx *ctx) {
    if (ctx->tile_blocks) return PM_OK;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
    int per_cu_t = 0, per_cu_q = 0, per_cu_p = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_t, k_gather_tile<0>, 64, 0));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_q, k_gather<0>, 256, 0));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_p, k_gather_pair<0>, 256, 0));
    int cus = std::max(8, prop.multiProcessorCount);
    ctx->tile_blocks = std::max(8, std::max(1, per_cu_t) * cus / 8 * 8);
    ctx->query_blocks = std::max(8, std::max(1, per_cu_q) * cus / 8 * 8);
    ctx->pair_blocks = std::max(8, std::max(1, per_cu_p) * cus / 8 * 8);
    if (const char *v = getenv("PM_TILE_BLOCKS")) { int x = atoi(v); if (x >= 8 && x <= ctx->tile_blocks) ctx->tile_blocks = x / 8 * 8; }   // experiments: fewer resident groups
    if (const char *v = getenv("PM_PAIR")) ctx->use_pairs = v[0] != '0';
    const char *e = getenv("PM_GATHER_TILE");
    ctx->use_tiles = !(e && e[0] == '0');
    const char *eb = getenv("PM_GROUP_BOUNCED");
    ctx->group_bounced = !(eb && eb[0] == '0');
    if (const char *v = getenv("PM_SP_CELL")) { double x = atof(v); if (x > 0.1 && x < 100) ctx->sp_cell = x; }
    if (const char *v = getenv("PM_SP_EST")) { double x = atof(v); if (x > 0.01 && x < 1000) ctx->sp_est = x; }
    if (const char *v = getenv("PM_TILE_PARTS")) { int x = atoi(v); if (x >= 1 && x <= 4) ctx->tile_parts = x; }
    if (const char *v = getenv("PM_LOOSE_TILES")) { double x = atof(v); if (x >= 0 && x < 1e6) { ctx->loose_tiles = x; ctx->loose_auto = false; } }
    if (const char *v = getenv("PM_SP_MIN_RUN")) { int x = atoi(v); if (x >= 1 && x <= 64) ctx->sp_min_run = x; }
    if (ctx->debug) fprintf(stderr, "[pm] resident gather blocks: group kernel %d (%d per CU), per-query kernel %d (%d per CU), %d CUs\n", ctx->tile_blocks, per_cu_t,
                                    ctx->query_blocks, per_cu_q, cus);
    return PM_OK;
}

static int render_impl(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, uint32_t *pane, int32_t pane_on_device, pm_stats *stats);
extern "C" int pm_render(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, uint32_t *pane, int32_t pane_on_device,
                         pm_stats *stats) {
    const int rc = render_impl(ctx, gmap, cmap, rp, pane, pane_on_device, stats);
    if (rc != PM_OK && ctx) {
        // a frame that fails half way may have gathers queued on the two streams beside the caller's (a map that was fine is
        // gathered before the other one's failure shows): nothing of it is left running when the error is reported, so the
        // caller may destroy or rebuild the maps at once
        const std::string msg = g_err;
        (void)hipSetDevice(ctx->device);
        for (int i = 0; i < 2; i++) if (ctx->slot[i].stream) (void)hipStreamSynchronize(ctx->slot[i].stream);
        if (ctx->sp_stream) (void)hipStreamSynchronize(ctx->sp_stream);
        (void)hipStreamSynchronize(ctx->stream);
        g_err = msg;
    }
    return rc;
}
// One frame, queued without a host wait: every count a later kernel needs (shade points, bounced pixels, groups, what the
// group kernel leaves) stays on the device, launches are sized for the capacity of the queues, and the maps are waited for by
// the streams (hipStreamWaitEvent), not by the host.  The host looks at the frame once, when it is over (render_impl).
static uint32_t left_of(const Counters &c, int kind) { return c.n_left[kind][0] + c.n_left[kind][1] + c.n_left[kind][2] + c.n_left[kind][3]; }
static int enqueue_frame(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, const FrameParams &fp, uint32_t *d_pane, int owned, bool gathered[2]) {
    const int W = rp->width;
    Counters *d_ctr = ctx->ctr.as<Counters>();
    const dim3 grid_px((W + 15) / 16, owned), block(256);
    const SPQueue q = sp_queue(ctx);
    NodeEntry *tab = ctx->tab.as<NodeEntry>();
    const uint32_t c_cap = ctx->c_cap, b_cap = ctx->b_cap;
    const unsigned cus8 = 256 * 8;

    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    HIPCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), ctx->stream));
    const bool count_tests = (rp->flags & PM_FLAG_COUNT_TESTS) != 0;
    // tiles spread out in depth are not groups: their points are sorted by where they lie, with the bounced pixels' points.
    // A tile of 8 pixels spans 8 / (fov |cam.z|) of its distance; `loose_tiles` of those spans is as deep as a group may be.
    const bool sort_bounced = ctx->use_tiles && ctx->group_bounced;
    double loose_k = 0.0;
    if (sort_bounced && ctx->loose_tiles > 0 && (ctx->loose_on || !ctx->loose_auto) && fp.cam.z != 0) {
        const double span = 8.0 / ((fp.width / 1280.0) * 64.0 * fabs(fp.cam.z));
        loose_k = (1.0 + ctx->loose_tiles * span) * (1.0 + ctx->loose_tiles * span);
    }
    if (count_tests) LAUNCH_SCENE(k_primary_count, grid_px, block, ctx, fp, d_pane, d_ctr, q, ctx->clist.as<uint32_t>(), ctx->groups.as<uint2>(), ctx->sp_loose.as<uint32_t>(), loose_k);
    else LAUNCH_SCENE(k_primary, grid_px, block, ctx, fp, d_pane, d_ctr, q, ctx->clist.as<uint32_t>(), ctx->groups.as<uint2>(), ctx->sp_loose.as<uint32_t>(), loose_k);
    hipLaunchKernelGGL(k_mark_primary, dim3(1), dim3(64), 0, ctx->stream, d_ctr, q.cap);
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    LAUNCH_SCENE(k_secondary, dim3(std::min<unsigned>((c_cap + 255) / 256, 4096u)), block, ctx, fp, d_ctr, q, ctx->clist.as<uint32_t>(), c_cap, tab);
    hipLaunchKernelGGL(k_mark_secondary, dim3(1), dim3(64), 0, ctx->stream, d_ctr, q.cap, b_cap);
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    // The shade points of bounced pixels (slots n_sp0 .. n_sp0 + n_bounced) in spatial order, for the group kernel: a chain of
    // thirteen small launches that needs the bounce pass and nothing else.  It runs beside the shadow pass on a stream of its own
    // (round 3; PM_SP_STREAM=0: behind the shadow pass on the caller's stream) and beside the maps' builds.
    auto queue_sp_order = [&](hipStream_t sps) -> int {
        HIPCHK(hipMemsetAsync(ctx->sp_bounds.p, 0xFF, 12, sps));                                   // minima
        HIPCHK(hipMemsetAsync(ctx->sp_bounds.as<unsigned char>() + 12, 0, 12, sps));             // maxima
        hipLaunchKernelGGL(k_sp_bounds, dim3(std::min(256u, (b_cap + 255) / 256)), block, 0, sps, q, d_ctr, ctx->sp_loose.as<uint32_t>(), ctx->sp_bounds.as<uint32_t>());
        hipLaunchKernelGGL(k_sp_morton, dim3((b_cap + 255) / 256), block, 0, sps, q, d_ctr, ctx->sp_loose.as<uint32_t>(), ctx->sp_bounds.as<uint32_t>(),
                           ctx->sp_keys_in.as<uint64_t>(), ctx->sp_vals_in.as<uint32_t>());
        HIPCHK(radix_sort_pairs_30(sps, ctx->sp_keys_in.as<uint64_t>(), ctx->sp_vals_in.as<uint32_t>(), ctx->sp_keys_out.as<uint64_t>(),
                                   ctx->sp_vals_out.as<uint32_t>(), ctx->sp_sort_tmp.as<uint32_t>(), b_cap, &d_ctr->n_sorted));
        // cells of about six 63-NN radii (measured best on the default scene; most runs are then left to the tile kernel's own
        // test of how far a group is spread): r^2 = 63 / (pi * density), density = photons / surface, surface taken as that
        // of a cube over the points' extent (the sparser of the two maps decides)
        long long n_ph = std::max<long long>(1, std::min(gmap ? gmap->n : (1LL << 40), cmap ? cmap->n : (1LL << 40)));
        double cell_rel = ctx->sp_cell * sqrt(63.0 * 6.0 / (3.14159265 * (double)n_ph));
        hipLaunchKernelGGL(k_sp_pieces, dim3((b_cap + 255) / 256), block, 0, sps, ctx->sp_keys_out.as<uint64_t>(), cell_rel, ctx->sp_pieces.as<uint2>(), d_ctr);
        // one wave per piece (a scene of many small objects has more surface than the cube over its points, which the estimate
        // of the photon density assumes: twice the expected set is allowed there; measured 8.55 -> 8.3 ms on 10 000 spheres)
        hipLaunchKernelGGL(k_sp_classify, dim3(std::min<unsigned>((b_cap + 3) / 4, 1024u)), block, 0, sps, q, ctx->sp_vals_out.as<uint32_t>(), ctx->sp_pieces.as<uint2>(),
                           ctx->sp_bounds.as<uint32_t>(), (double)n_ph / 6.0, ctx->sp_est * (loose_k > 0 && ctx->loose_auto ? 2.0 : 1.0) * (double)PM_TILE_EST_MAX, (uint32_t)ctx->sp_min_run,
                           ctx->sp_groups.as<uint2>(), d_ctr, ctx->sp_early.as<uint32_t>());
        HIPCHK(hipGetLastError());
        return PM_OK;
    };
    if (sort_bounced && ctx->sp_on_side) {
        HIPCHK(hipEventRecord(ctx->ev_sec, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->sp_stream, ctx->ev_sec, 0));
        PMCHK(queue_sp_order(ctx->sp_stream));
        HIPCHK(hipEventRecord(ctx->ev_sp, ctx->sp_stream));
    }
    // one lane per shadow ray; or (A/B switch) one lane per shade point with its 25 rays in sequence
    const dim3 grid_sh(std::min<unsigned>((q.cap + 9) / 10, 4 * cus8));
    if (count_tests && ctx->shadow_queue) LAUNCH_SCENE(k_shadow_rays_count, grid_sh, block, ctx, fp, q, d_ctr);
    else if (count_tests) LAUNCH_SCENE(k_shadow_rays_lane_count, grid_sh, block, ctx, fp, q, d_ctr);
    else if (ctx->shadow_per_ray && ctx->shadow_queue) LAUNCH_SCENE(k_shadow_rays, grid_sh, block, ctx, fp, q, d_ctr);
    else if (ctx->shadow_per_ray) LAUNCH_SCENE(k_shadow_rays_lane, grid_sh, block, ctx, fp, q, d_ctr);
    else LAUNCH_SCENE(k_shadow, dim3(std::min<unsigned>((q.cap + 255) / 256, 2 * cus8)), block, ctx, fp, q, d_ctr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    // The gather: groups of neighbouring shade points first (k_gather_tile), then one wave per shade point for what
    // the group kernel left and for the shade points of bounced pixels (k_gather).  Both kernels are launched with
    // as many blocks as the device keeps resident and draw their work from device counters.
    HIPCHK(hipMemsetAsync(ctx->sched.p, 0, N_SCHED * SCHED_WORDS * 4, ctx->stream));   // 64 work counters per gather launch
    int n_sched = 0;
    uint32_t *sched = ctx->sched.as<uint32_t>();
    uint32_t *left = ctx->left.as<uint32_t>();
    const bool use_tiles = ctx->use_tiles;
    const int dbg = ctx->debug ? 1 : 0;
    if (sort_bounced) {
        if (ctx->sp_on_side) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sp, 0));     // (queued beside the shadow pass, above)
        else PMCHK(queue_sp_order(ctx->stream));
    }
    const uint32_t *sorted = ctx->sp_vals_out.as<uint32_t>();
    // what the group kernel does not take is the per-query kernel's range: everything without the group kernel, the
    // bounced pixels' shade points when they are not grouped, else nothing but what the group kernel leaves
    const int sec_mode = use_tiles ? (sort_bounced ? 0 : 1) : 2;
    // Streams side by side, so that the thin end of one launch (few items left, most CUs idle) is filled by the others:
    // the global map's groups on the caller's stream, the caustic map's groups on a second; each group kernel is followed,
    // on its stream, by the per-query kernel for what it left.  A third stream carries the per-query gather of the bounced
    // pixels' shade points that are not grouped.
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    HIPCHK(hipStreamWaitEvent(ctx->aux2, ctx->ev_fork, 0));
    const uint32_t *zero = &d_ctr->pad;   // "nothing left": the third stream's launches have only the bounced range
    // the map whose build is already over goes first
    const bool global_late = gmap && gmap->pending && hipEventQuery(gmap->ready) != hipSuccess;
    const uint32_t nparts = use_tiles ? (uint32_t)ctx->tile_parts : 1u;
    auto per_query = [&](int kind, pm_map *map, hipStream_t on, const uint32_t *list, const uint32_t *count, int mode) {
        uint32_t *sc = sched + (size_t)(n_sched++) * SCHED_WORDS;
        if (ctx->use_pairs) {
            if (kind == 0) hipLaunchKernelGGL(k_gather_pair<0>, dim3(ctx->pair_blocks), block, 0, on, map->view, q, list, count, d_ctr, mode, sc);
            else hipLaunchKernelGGL(k_gather_pair<1>, dim3(ctx->pair_blocks), block, 0, on, map->view, q, list, count, d_ctr, mode, sc);
        } else if (kind == 0) hipLaunchKernelGGL(k_gather<0>, dim3(ctx->query_blocks), block, 0, on, map->view, q, list, count, d_ctr, mode, sc);
        else hipLaunchKernelGGL(k_gather<1>, dim3(ctx->query_blocks), block, 0, on, map->view, q, list, count, d_ctr, mode, sc);
    };
    // first the waits for the maps, and on the third stream what needs no group kernel before it
    for (int turn = 0; turn < 2; turn++) {
        const int kind = global_late ? 1 - turn : turn;
        pm_map *map = kind == 0 ? gmap : cmap;
        if (rp->flags & (kind == 0 ? PM_FLAG_NO_GLOBAL : PM_FLAG_NO_CAUSTIC)) continue;
        hipStream_t st = kind == 0 ? ctx->stream : ctx->aux;
        // the map may still be under construction on its build stream (beside the passes above, and beside the other
        // map's gather): the gather's streams wait for it, the host does not
        if (map->failed != PM_OK) return fail(map->failed, map->fail_msg);
        if (map->pending) {
            if (!ctx->yield_known[map->kind >= 0 ? map->kind : 0] && map->kind >= 0) PMCHK(map_ready(map));   // first map of a scene: is it complete at all?
            else { HIPCHK(hipStreamWaitEvent(st, map->ready, 0)); HIPCHK(hipStreamWaitEvent(ctx->aux2, map->ready, 0)); }
        }
        HIPCHK(hipEventRecord(ctx->ev_g[kind][0], st));
        if (use_tiles) {
            // beside the groups: the bounced pixels' shade points that are not grouped
            if (sec_mode == 1) per_query(kind, map, ctx->aux2, left, zero, 1);
            else per_query(kind, map, ctx->aux2, ctx->sp_early.as<uint32_t>(), &d_ctr->n_early, 0);
        }
        gathered[kind] = true;
    }
    // then the groups, part by part; what a part leaves is gathered per query on the third stream beside the next part, what
    // the last part leaves behind it on its own stream
    for (uint32_t part = 0; part < nparts; part++) {
        for (int turn = 0; turn < 2; turn++) {
            const int kind = global_late ? 1 - turn : turn;
            pm_map *map = kind == 0 ? gmap : cmap;
            if (!gathered[kind]) continue;
            hipStream_t st = kind == 0 ? ctx->stream : ctx->aux;
            float4 *strips = ctx->strips.as<float4>() + (size_t)kind * ctx->tile_blocks * TS_STRIDE;
            uint32_t *lk = left + ((size_t)kind * nparts + part) * q.cap, *nl = &d_ctr->n_left[kind][part];
            if (use_tiles) {
                uint32_t *sc = sched + (size_t)(n_sched++) * SCHED_WORDS;
                if (kind == 0) hipLaunchKernelGGL(k_gather_tile<0>, dim3(ctx->tile_blocks), dim3(64), 0, st, map->view, q, ctx->groups.as<uint2>(), d_ctr, lk, nl, sc, strips, dbg,
                                                  sorted, ctx->sp_groups.as<uint2>(), part, nparts);
                else hipLaunchKernelGGL(k_gather_tile<1>, dim3(ctx->tile_blocks), dim3(64), 0, st, map->view, q, ctx->groups.as<uint2>(), d_ctr, lk, nl, sc, strips, dbg,
                                        sorted, ctx->sp_groups.as<uint2>(), part, nparts);
            }
            if (part + 1 < nparts) {
                HIPCHK(hipEventRecord(ctx->ev_part[kind][part], st));
                HIPCHK(hipStreamWaitEvent(ctx->aux2, ctx->ev_part[kind][part], 0));
                per_query(kind, map, ctx->aux2, lk, nl, 0);
            } else {
                HIPCHK(hipEventRecord(ctx->ev_g[kind][1], st));
                per_query(kind, map, st, lk, nl, use_tiles ? 0 : 2);
                HIPCHK(hipEventRecord(ctx->ev_g[kind][2], st));
            }
        }
    }
    HIPCHK(hipEventRecord(ctx->ev_join2, ctx->aux2));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join2, 0));
    HIPCHK(hipEventRecord(ctx->ev[4], ctx->stream));            // the global gather is done
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->aux));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    HIPCHK(hipEventRecord(ctx->ev[5], ctx->stream));            // both are done
    LAUNCH_SCENE(k_shade, dim3(std::min<unsigned>((q.cap + 255) / 256, 2 * cus8)), block, ctx, fp, q, d_ctr, d_pane, tab);
    LAUNCH_SCENE(k_resolve, dim3(std::min<unsigned>((c_cap + 255) / 256, cus8)), block, ctx, ctx->clist.as<uint32_t>(), d_ctr, c_cap, tab, d_pane);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[6], ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
    // the end of this frame, for pm_map_destroy: the caller's stream has joined the other two above
    ctx->frame_seq++;
    HIPCHK(hipEventRecord(ctx->ev_frame, ctx->stream));
    if (gathered[0] && gmap) gmap->used_in_frame = ctx->frame_seq;
    if (gathered[1] && cmap) cmap->used_in_frame = ctx->frame_seq;
    return PM_OK;
}

static int render_impl(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, uint32_t *pane, int32_t pane_on_device, pm_stats *stats) {
    FrameParams fp;
    PMCHK(make_frame_params(ctx, rp, fp));
    if (!pane) return fail(PM_EINVAL, "pm_render: pane is NULL");
    if (!(rp->flags & PM_FLAG_NO_GLOBAL) && !gmap) return fail(PM_EINVAL, "pm_render: global map missing (or pass PM_FLAG_NO_GLOBAL)");
    if (!(rp->flags & PM_FLAG_NO_CAUSTIC) && !cmap) return fail(PM_EINVAL, "pm_render: caustic map missing (or pass PM_FLAG_NO_CAUSTIC)");
    if ((gmap && gmap->ctx != ctx) || (cmap && cmap->ctx != ctx)) return fail(PM_EINVAL, "pm_render: map belongs to another context");
    PMCHK(set_device(ctx));
    const int W = rp->width, H = rp->height;
    const int nblk = (H + PM_ROW_BLOCK - 1) / PM_ROW_BLOCK;
    const int owned = nblk > rp->rank ? (nblk - rp->rank + rp->world - 1) / rp->world : 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (owned == 0) return PM_OK;
    const size_t owned_px = (size_t)owned * PM_ROW_BLOCK * W;
    if (owned_px + owned_px / 4 > 0xFFFFFF00u) return fail(PM_EINVAL, "pm_render: too many pixels for one call; split the frame over more ranks");

    uint32_t *d_pane = pane;
    if (!pane_on_device) { PMCHK(ctx->pane.ensure((size_t)W * H * 4)); d_pane = ctx->pane.as<uint32_t>(); }
    PMCHK(ctx->ctr.ensure(sizeof(Counters)));
    PMCHK(ctx->clist.ensure(owned_px * 4));
    PMCHK(ctx->groups.ensure((size_t)owned * 2 * ((W + 7) / 8) * sizeof(uint2)));   // one group per 8x8 pixel tile: 2 tile rows per row block, ceil(W/8) columns
    PMCHK(gather_blocks(ctx));
    PMCHK(ctx->sched.ensure(N_SCHED * SCHED_WORDS * 4 + 2 * 8192 * 16));
    PMCHK(ctx->strips.ensure((size_t)2 * ctx->tile_blocks * TS_STRIDE * sizeof(float4)));
    PMCHK(ctx->sp_bounds.ensure(6 * 4));
    // Capacities the frame is launched for.  They start from what frames of this size usually need and only grow: a frame
    // that needed more says so (Counters::overflow) and is rendered again with what it asked for.
    uint32_t want_sp = (uint32_t)(owned_px + owned_px / 4), want_c = (uint32_t)(owned_px / 8 + 4096), want_b = (uint32_t)(owned_px / 8 + 4096);
    bool gathered[2] = {false, false};
    for (int attempt = 0;; attempt++) {
        PMCHK(ensure_sp(ctx, want_sp));
        PMCHK(ctx->left.ensure((size_t)2 * ctx->tile_parts * ctx->sp_cap * 4 + 8));   // per map and part of the group launches
        PMCHK(ctx->sp_loose.ensure((size_t)ctx->sp_cap * 4));
        if (want_c > ctx->c_cap) { PMCHK(ctx->tab.ensure((size_t)want_c * 16 * sizeof(NodeEntry))); ctx->c_cap = want_c; }
        if (want_b > ctx->b_cap) {
            PMCHK(ctx->sp_keys_in.ensure((size_t)want_b * 8)); PMCHK(ctx->sp_keys_out.ensure((size_t)want_b * 8));
            PMCHK(ctx->sp_vals_in.ensure((size_t)want_b * 4)); PMCHK(ctx->sp_vals_out.ensure((size_t)want_b * 4));
            PMCHK(ctx->sp_sort_tmp.ensure(sort_scratch_words(want_b) * 4));
            PMCHK(ctx->sp_groups.ensure((size_t)want_b * sizeof(uint2))); PMCHK(ctx->sp_early.ensure((size_t)want_b * 4)); PMCHK(ctx->sp_pieces.ensure((size_t)want_b * sizeof(uint2)));
            ctx->b_cap = want_b;
        }
        gathered[0] = gathered[1] = false;
        PMCHK(enqueue_frame(ctx, gmap, cmap, rp, fp, d_pane, owned, gathered));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->frame_waited = ctx->frame_seq;
        const Counters &hc = *ctx->h_ctr;
        bool again = false;
        if (hc.overflow) {
            if (ctx->sp_cap_limit > 0 || attempt >= 3)
                return fail(PM_ESTATE, "pm_render: the shade point queue overflowed (" + std::to_string(hc.n_sp) + " shade points): the pane is incomplete");
            if (hc.n_sp > 0xFFFFFF00u) return fail(PM_EINVAL, "pm_render: too many shade points for one call; split the frame over more ranks");
            // the counters kept counting past the capacities: they say what the frame needs
            if (hc.overflow & OVF_SP) want_sp = std::max(want_sp, hc.n_sp);
            if (hc.overflow & OVF_COMPLEX) { want_c = hc.n_complex + hc.n_complex / 8; want_sp = std::max<uint64_t>(want_sp, std::min<uint64_t>((uint64_t)hc.n_sp + 15ull * (hc.n_complex - ctx->c_cap), 0xFFFFFF00u)); }
            if (hc.overflow & OVF_BOUNCED) { const uint32_t need = (hc.n_sp - hc.n_sp0) + hc.n_loose; want_b = need + need / 8; }
            again = true;
        }
        // the maps the streams waited for: did their one emission range hold n photons?  (a scene that stores far fewer photons
        // per attempt than the last map of this scene did: topped up and indexed again here, and the frame is rendered again)
        for (int kind = 0; kind < 2; kind++) {
            pm_map *map = kind == 0 ? gmap : cmap;
            if (!gathered[kind] || !map) continue;
            bool rebuilt = false;
            PMCHK(map_ready(map, &rebuilt));
            again = again || rebuilt;
        }
        if (!again) break;
    }
    if (!pane_on_device && rp->world == 1) {
        HIPCHK(hipMemcpyAsync(pane, d_pane, (size_t)W * H * 4, hipMemcpyDeviceToHost, ctx->stream));   // one rank owns every row: one copy
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else if (!pane_on_device) {
        for (int b = rp->rank; b < nblk; b += rp->world) {
            size_t y0 = (size_t)b * PM_ROW_BLOCK, y1 = std::min<size_t>(y0 + PM_ROW_BLOCK, (size_t)H);
            HIPCHK(hipMemcpyAsync(pane + y0 * W, d_pane + y0 * W, (y1 - y0) * W * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    const Counters hc = *ctx->h_ctr;
    if (ctx->loose_auto && ctx->use_tiles && hc.n_sp0 > 0) {
        // the next frame of this scene: tiles that are deep go through the spatial sort when this frame handed a good part of its
        // shade points over to the per-query kernel (groups too spread out for one shared set, mostly; both maps counted)
        const int n_maps = (gathered[0] ? 1 : 0) + (gathered[1] ? 1 : 0);
        if (!ctx->loose_on) ctx->loose_on = n_maps > 0 && (double)left_of(hc, 0) + (double)left_of(hc, 1) > 0.15 * n_maps * (double)hc.n_sp0;
        else ctx->loose_on = (double)hc.n_loose > 0.05 * (double)hc.n_sp0;
    }
    const uint32_t n_sp0 = hc.n_sp0, n_sp = hc.n_sp0 + hc.n_bounced, n_complex = hc.n_complex, n_secondary = hc.n_secondary;
    const bool sort_bounced = ctx->use_tiles && ctx->group_bounced;
    const uint32_t n_sorted = sort_bounced ? hc.n_bounced : 0u, n_primary_sp = ctx->use_tiles ? n_sp0 + n_sorted : 0u;
    const int n_gathers = (gathered[0] ? 1 : 0) + (gathered[1] ? 1 : 0);
    Counters *d_ctr = ctx->ctr.as<Counters>();
    if (ctx->debug) {
        fprintf(stderr, "[pm] shade points %u (primary %u), groups %u, left to the per-query gather: global %u caustic %u\n", n_sp, n_sp0,
                hc.n_groups, left_of(hc, 0), left_of(hc, 1));
        fprintf(stderr, "[pm] shade points in spatial order: %u of bounced pixels + %u of spread-out tiles, in %u groups; %u straight to the per-query kernel\n", n_sorted,
                hc.n_sorted - hc.n_bounced, hc.n_sgroups, hc.n_early);
        fprintf(stderr, "[pm] lanes handed over, by reason (few photons, spread, radius, |S|, no lane served, -, own shortfall): %u %u %u %u %u %u %u\n", hc.why[1], hc.why[2],
                hc.why[3], hc.why[4], hc.why[5], hc.why[6], hc.why[7]);
        for (int k = 0; k < 2; k++) {
            fprintf(stderr, "[pm] groups of map %d by log2(ticks):", k);
            for (int b = 10; b < 32; b++) fprintf(stderr, " %d:%u", b, hc.ghist[k][b]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[pm] group kernel ticks per phase (centre query, collect, pass 1, pass 2, rank, sum): %llu %llu %llu %llu %llu %llu; shared sets %llu, candidates %llu\n", hc.tphase[0],
                hc.tphase[1], hc.tphase[2], hc.tphase[3], hc.tphase[4], hc.tphase[5], hc.tphase[7], hc.tphase[6]);
#ifdef PM_PAIR_STATS
        fprintf(stderr, "[pm] per-query gathers answered by the half-wave block query: global %u of %u, caustic %u of %u\n", hc.ghist[0][1], hc.ghist[0][0], hc.ghist[1][1], hc.ghist[1][0]);
#endif
        if (hc.walk[0]) fprintf(stderr, "[pm] shadow pass through the sphere grid: %llu sphere tests, %llu cells entered by %u rays; the waves made %llu steps and %llu rounds of tests\n",
                                hc.tests[2], hc.walk[0], 25u * n_sp, hc.walk[1], hc.walk[2]);
    }
    (void)d_ctr;
#ifdef PM_WAVE_TIMES
    if (const char *path = getenv("PM_WAVE_DUMP")) {
        std::vector<uint32_t> h(2 * 8192 * 4);
        if (n_sp && hipMemcpy(h.data(), ctx->sched.as<uint32_t>() + 6 * SCHED_WORDS, h.size() * 4, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
    }
#endif
    if (stats) {
        size_t px = 0;
        for (int b = rp->rank; b < nblk; b += rp->world) px += (std::min<size_t>((size_t)(b + 1) * PM_ROW_BLOCK, (size_t)H) - (size_t)b * PM_ROW_BLOCK) * W;
        stats->closest_hit_queries = px + n_secondary;
        stats->shadow_queries = (uint64_t)25 * n_sp;
        stats->gathers = (uint64_t)n_gathers * n_sp;
        stats->absorb_points = n_sp;
        stats->complex_pixels = n_complex;
        stats->tests_primary_sphere = hc.tests[0]; stats->tests_primary_plane = hc.tests[1];
        stats->tests_shadow_sphere = hc.tests[2]; stats->tests_shadow_plane = hc.tests[3];
        (void)hipEventElapsedTime(&stats->ms_primary, ctx->ev[0], ctx->ev[1]);
        (void)hipEventElapsedTime(&stats->ms_secondary, ctx->ev[1], ctx->ev[2]);
        (void)hipEventElapsedTime(&stats->ms_shadow, ctx->ev[2], ctx->ev[3]);
        (void)hipEventElapsedTime(&stats->ms_gather, ctx->ev[3], ctx->ev[5]);
        if (gathered[0]) {
            (void)hipEventElapsedTime(&stats->ms_gather_global, ctx->ev_g[0][0], ctx->ev_g[0][2]);
            (void)hipEventElapsedTime(&stats->ms_group_global, ctx->ev_g[0][0], ctx->ev_g[0][1]);
            stats->grouped_global = n_primary_sp ? n_primary_sp - left_of(hc, 0) - hc.n_early : 0u;
        }
        if (gathered[1]) {
            (void)hipEventElapsedTime(&stats->ms_gather_caustic, ctx->ev_g[1][0], ctx->ev_g[1][2]);
            (void)hipEventElapsedTime(&stats->ms_group_caustic, ctx->ev_g[1][0], ctx->ev_g[1][1]);
            stats->grouped_caustic = n_primary_sp ? n_primary_sp - left_of(hc, 1) - hc.n_early : 0u;
        }
        stats->left_global = gathered[0] ? left_of(hc, 0) : 0u; stats->left_caustic = gathered[1] ? left_of(hc, 1) : 0u;
        stats->early_points = ctx->use_tiles ? (sort_bounced ? hc.n_early : hc.n_bounced) : 0u;
        (void)hipEventElapsedTime(&stats->ms_resolve, ctx->ev[5], ctx->ev[6]);
        (void)hipEventElapsedTime(&stats->ms_total, ctx->ev[0], ctx->ev[6]);
    }
    return PM_OK;
}

// raytrace::render between scene set-up and the PNG writer (src/Image.cpp:32-57): the supersampled pane stays on the
// device, only the width x height image travels.
extern "C" int pm_render_image(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, int32_t sample_ratio, int32_t flip, uint32_t *out,
                               int32_t out_on_device, pm_stats *stats) {
    if (!ctx || !rp || !out) return fail(PM_EINVAL, "pm_render_image: NULL argument");
    if (sample_ratio < 1 || rp->width % sample_ratio || rp->height % sample_ratio)
        return fail(PM_EINVAL, "pm_render_image: the pane (rp->width x rp->height) must be sample_ratio times the image in both directions");
    if (rp->world != 1) return fail(PM_EINVAL, "pm_render_image: one rank renders the whole pane (world must be 1)");
    PMCHK(set_device(ctx));
    PMCHK(ctx->pane.ensure((size_t)rp->width * rp->height * 4));
    PMCHK(pm_render(ctx, gmap, cmap, rp, ctx->pane.as<uint32_t>(), 1, stats));
    return pm_downsample(ctx, ctx->pane.as<uint32_t>(), 1, rp->width / sample_ratio, rp->height / sample_ratio, sample_ratio, flip, out, out_on_device);
}

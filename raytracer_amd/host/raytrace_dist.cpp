// raytrace_dist.cpp -- the one-process-per-GPU frame of raytrace_dist.h: RCCL directly, no Python.
// Every photon and every pixel still comes from libraytracer_amd.so; this file only moves shards and rows between ranks.
#include "raytrace_dist.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "raytrace_backend.h"

namespace raytrace {

namespace {
constexpr int kRowBlock = PM_ROW_BLOCK;
constexpr uint64_t kProbeAttempts = 65536;

void hip_ok(hipError_t e, const char *what) {
    if (e != hipSuccess) { std::fprintf(stderr, "raytracer_amd: %s failed: %s\n", what, hipGetErrorString(e)); std::abort(); }
}
void nccl_ok(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) { std::fprintf(stderr, "raytracer_amd: %s failed: %s\n", what, ncclGetErrorString(r)); std::abort(); }
}
struct DeviceBuffer {            // grow-only
    void *p = nullptr;
    size_t bytes = 0;
    void ensure(size_t want) {
        if (want <= bytes) return;
        if (p) hip_ok(hipFree(p), "hipFree");
        hip_ok(hipMalloc(&p, want), "hipMalloc");
        bytes = want;
    }
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
};
}  // namespace

struct RankGroup {
    int32 rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DeviceBuffer mine[2], every[2], pane, probe;
    double yield[2] = {0, 0};        // stored photons per attempt of the uploaded scene, probed once per scene
    std::vector<unsigned char> probed_for;
    std::string id_file;
};

RankGroup *joinRanks(int32 rank, int32 world, const char *id_file) {
    if (world < 1 || rank < 0 || rank >= world || world > 64 || !id_file) { std::fprintf(stderr, "raytracer_amd: joinRanks needs 0 <= rank < world <= 64 and an id file\n"); std::abort(); }
    RankGroup *g = new RankGroup();
    g->rank = rank; g->world = world; g->id_file = id_file;
    pm_ctx *ctx = backend::context();            // before RCCL: the library's streams get hardware queues of their own
    hip_ok(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking), "hipStreamCreate");
    backend::checked(pm_ctx_set_stream(ctx, g->stream), "pm_ctx_set_stream");
    // RCCL's unique id travels through a file: {token, id}.  The token (RAYTRACER_GROUP_TOKEN, or --token of the CLI: any
    // number the launcher gives every rank of ONE launch, e.g. its own pid) tells this launch's file from one that a rank 0
    // which died left behind: the other ranks wait for a file with THEIR token instead of taking a stale id and hanging in
    // ncclCommInitRank.  Without a token (0) any file is taken, as before: the launcher then has to remove stale files itself.
    unsigned long long token = 0;
    if (const char *e = std::getenv("RAYTRACER_GROUP_TOKEN")) token = std::strtoull(e, nullptr, 0);
    struct IdRecord { unsigned long long token; ncclUniqueId id; } rec;
    std::memset(&rec, 0, sizeof(rec));
    ncclUniqueId id;
    if (rank == 0) {
        nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
        rec.token = token; rec.id = id;
        const std::string tmp = std::string(id_file) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&rec, sizeof(rec), 1, f) != 1) { std::fprintf(stderr, "raytracer_amd: cannot write %s\n", tmp.c_str()); std::abort(); }
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file) != 0) { std::fprintf(stderr, "raytracer_amd: cannot create %s\n", id_file); std::abort(); }
    } else {
        bool got = false;
        for (int tries = 0; tries < 1200 && !got; tries++) {      // up to two minutes for rank 0 to come up
            if (FILE *f = std::fopen(id_file, "rb")) { got = std::fread(&rec, sizeof(rec), 1, f) == 1 && rec.token == token; std::fclose(f); }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) { std::fprintf(stderr, "raytracer_amd: rank %d never saw %s with this launch's token\n", rank, id_file); std::abort(); }
        id = rec.id;
    }
    nccl_ok(ncclCommInitRank(&g->comm, world, id, rank), "ncclCommInitRank");
    return g;
}

void leaveRanks(RankGroup *g) {
    if (!g) return;
    backend::checked(pm_ctx_synchronize(backend::context()), "pm_ctx_synchronize");
    backend::checked(pm_ctx_set_stream(backend::context(), nullptr), "pm_ctx_set_stream");
    if (g->comm) (void)ncclCommDestroy(g->comm);
    if (g->rank == 0) (void)std::remove(g->id_file.c_str());      // every rank has long read it: the next group starts from a fresh file
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

namespace {
// every rank finds the same numbers: attempts 0 .. 65535 of each map kind, less a twentieth (raytracer_amd/frame.py: yields)
void probe_yields(RankGroup *g, Scene *scene, uint64_t seed, const double lc[3]) {
    std::vector<unsigned char> key((size_t)scene->size * sizeof(pm_prim));
    for (int i = 0; i < scene->size; i++) { pm_prim p; std::memset(&p, 0, sizeof(p)); scene->objects[i]->flatten(&p); std::memcpy(&key[(size_t)i * sizeof(pm_prim)], &p, sizeof(p)); }
    if (key == g->probed_for && g->yield[0] > 0) return;
    g->probe.ensure(kProbeAttempts * sizeof(pm_photon));
    for (int kind = 0; kind < 2; kind++) {
        uint64_t got = 0;
        backend::checked(pm_emit_range(backend::context(), kind, seed, lc, 0, kProbeAttempts, static_cast<pm_photon *>(g->probe.p), kProbeAttempts, &got), "pm_emit_range");
        g->yield[kind] = std::max(0.95 * (double)got / (double)kProbeAttempts, 1e-4);
    }
    g->probed_for = key;
}
pm_map *map_from_shards(RankGroup *g, int kind, int64_t n, uint64_t seed, const double lc[3]) {
    pm_ctx *ctx = backend::context();
    const uint64_t total = (uint64_t)((double)n / g->yield[kind]) + 4096ull * (uint64_t)g->world;
    const uint64_t per = (total + (uint64_t)g->world - 1) / (uint64_t)g->world;
    const uint64_t cap = (uint64_t)((double)per * std::min(1.0, g->yield[kind] / 0.95 + 0.10)) + 1024;
    const size_t nbytes = 32 + 32 * (size_t)cap;
    g->mine[kind].ensure(nbytes);
    g->every[kind].ensure(nbytes * (size_t)g->world);
    backend::checked(pm_emit_shard(ctx, kind, seed, lc, (uint64_t)g->rank * per, per, g->mine[kind].p, cap), "pm_emit_shard");
    nccl_ok(ncclAllGather(g->mine[kind].p, g->every[kind].p, nbytes, ncclUint8, g->comm, g->stream), "ncclAllGather");
    pm_map *map = nullptr;
    backend::checked(pm_map_build_shards(ctx, g->every[kind].p, g->world, nbytes, n, &map), "pm_map_build_shards");
    return map;
}
}  // namespace

namespace {
// the frame; when it returns, rank 0's device pane (g->pane) is complete and the group's stream is idle
void frame_on_ranks(RankGroup *g, Scene *scene, Vec3 &camera, int32 width, int32 height) {
    pm_ctx *ctx = backend::context();
    backend::uploadScene(scene);
    const uint64_t seed = backend::frameSeed();
    const double lc[3] = {0.6, 0.6, 0.6};                       // light_color of renderScene, src/Raytrace.cpp:303
    int32 counts[2];
    backend::photonCounts(&counts[0], &counts[1]);
    if (g->rank == 0) std::printf("Building global and caustic photon maps from %d + %d photons on %d ranks\n", counts[0], counts[1], g->world);
    probe_yields(g, scene, seed, lc);
    pm_map *maps[2] = {nullptr, nullptr};
    for (int kind = 1; kind >= 0; kind--) maps[kind] = map_from_shards(g, kind, counts[kind], seed, lc);   // the caustic map takes longer: first
    if (g->rank == 0) std::printf("Rendering scene\n");
    pm_render_params rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.width = width; rp.height = height;
    rp.cam[0] = camera.x; rp.cam[1] = camera.y; rp.cam[2] = camera.z;
    rp.light_color[0] = lc[0]; rp.light_color[1] = lc[1]; rp.light_color[2] = lc[2];
    rp.seed = seed; rp.rank = g->rank; rp.world = g->world;
    g->pane.ensure((size_t)width * (size_t)height * 4);
    uint32_t *d_pane = static_cast<uint32_t *>(g->pane.p);
    pm_stats st;
    int rc = pm_render(ctx, maps[0], maps[1], &rp, d_pane, 1, &st);
    if (rc == PM_ESTATE && std::strstr(pm_last_error(), "pm_map_build_shards")) {
        // the ranges did not hold n photons after all (every rank sees the same counts, so every rank lands here): every rank
        // emits both maps itself -- the random streams are counter-based, the maps are the same on all of them
        for (int kind = 0; kind < 2; kind++) {
            backend::checked(pm_map_destroy(maps[kind]), "pm_map_destroy");
            backend::checked(pm_map_create(ctx, kind, counts[kind], seed, lc, &maps[kind], nullptr), "pm_map_create");
        }
        rc = pm_render(ctx, maps[0], maps[1], &rp, d_pane, 1, &st);
    }
    backend::checked(rc, "pm_render");
    // rows: interleaved blocks of 16, block b traced by rank b mod world, sent to rank 0 straight into place
    const int nblk = (height + kRowBlock - 1) / kRowBlock;
    if (g->world > 1) {
        nccl_ok(ncclGroupStart(), "ncclGroupStart");
        for (int b = 0; b < nblk; b++) {
            const int owner = b % g->world;
            const size_t rows = (size_t)std::min(kRowBlock, height - b * kRowBlock), words = rows * (size_t)width;
            uint32_t *at = d_pane + (size_t)b * kRowBlock * (size_t)width;
            if (owner != 0 && g->rank == owner) nccl_ok(ncclSend(at, words, ncclUint32, 0, g->comm, g->stream), "ncclSend");
            if (owner != 0 && g->rank == 0) nccl_ok(ncclRecv(at, words, ncclUint32, owner, g->comm, g->stream), "ncclRecv");
        }
        nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    }
    hip_ok(hipStreamSynchronize(g->stream), "hipStreamSynchronize");
    for (int kind = 0; kind < 2; kind++) backend::checked(pm_map_destroy(maps[kind]), "pm_map_destroy");
    if (g->rank == 0) std::printf("Scene rendered: %llu closest-hit queries, %llu gathers on rank 0\n", (unsigned long long)st.closest_hit_queries, (unsigned long long)st.gathers);
}
}  // namespace

void renderSceneRanks(RankGroup *g, Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height) {
    frame_on_ranks(g, scene, camera, width, height);
    if (g->rank == 0) hip_ok(hipMemcpy(pane, g->pane.p, (size_t)width * (size_t)height * 4, hipMemcpyDeviceToHost), "hipMemcpy");
}

void renderRanks(RankGroup *g, const char *image_file, int32 width, int32 height, int32 sample_ratio, Scene *scene) {
    (void)backend::frameSeed();
    if (g->rank == 0) std::printf("Setting up scene\n");
    if (!scene) scene = defaultScene();
    const int s = sample_ratio;
    Vec3 camera(0, 0, -12);
    std::vector<uint32> image((size_t)width * height);
    if (width > 0 && height > 0 && s > 0) {
        frame_on_ranks(g, scene, camera, width * s, height * s);
        // the front-end of src/Image.cpp:36-57 on rank 0's GPU, where the supersampled pane is: box filter with integer
        // division, vertical flip; width x height words come back
        if (g->rank == 0) backend::checked(pm_downsample(backend::context(), static_cast<const uint32_t *>(g->pane.p), 1, width, height, s, 1, image.data(), 0), "pm_downsample");
    }
    delete scene;
    if (g->rank == 0 && !backend::writeImage(image_file, width, height, image.data())) std::fprintf(stderr, "raytracer_amd: cannot write %s\n", image_file);
}

}  // namespace raytrace

/*
 * raytracer_amd.h -- C-ABI of libraytracer_amd.so (MI355X / gfx950 HIP backend).
 *
 * The reference (Deamon5550/raytracer) has no FFI; its seam is the JobSystem worker pool
 * (src/JobSystem.h:5-19) plus the C++ functions the jobs call.  This library replaces the CPU
 * workers for the hot path only: per-pixel trace / intersect / shade (src/Raytrace.cpp:32-299)
 * and photon emission + k-NN radiance gather (src/PhotonMap.cpp).  Each entry point names the
 * reference interface it stands in for.  All functions return 0 on success, a negative PM_E*
 * code otherwise (the reference has no error channel: every function there returns void);
 * pm_last_error() gives the message.  One context per process and GPU; calls on one context
 * must come from one host thread at a time.  All work is issued on the context's stream
 * (default: the null stream, so it orders with PyTorch-ROCm's default stream).
 *
 * Random numbers follow "RNG contract v1" (DESIGN.md): counter-based streams keyed by
 * (seed, kind, item, draw#) that replace the reference's single racy global engine
 * (src/Random.cpp:8-27).
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_EINVAL (-1)   /* bad argument */
#define PM_EHIP (-2)     /* HIP runtime error */
#define PM_ENOMEM (-3)
#define PM_ESTATE (-4)   /* e.g. no scene uploaded */

/* Flattened SceneObject / SphereObject / PlaneObject (src/Scene.h:8-59); array index = hit id. */
typedef struct pm_prim {
    int32_t kind;               /* 0 = SphereObject, 1 = PlaneObject */
    int32_t pad0;
    double x, y, z;             /* SceneObject::x,y,z (plane: exactly one non-zero selects the axis) */
    double radius;              /* SphereObject::radius */
    double dx, dy, dz;          /* SphereObject::dx,dy,dz */
    double min_bound, max_bound;/* PlaneObject::min_bound,max_bound */
    float red, green, blue;     /* colour byte / 255.0f (src/Scene.cpp:54-56) */
    float pad1;
    double absorb_chance, diffuse_chance, specular_chance, transmission_chance;
    double refraction, specular_coeff;
} pm_prim;                      /* 144 bytes */

/* struct photon, src/PhotonMap.h:8-13 (32 bytes, same field order). */
typedef struct pm_photon {
    float x, y, z;
    uint8_t power[4];
    float dx, dy, dz;
    int8_t bounce;
    uint8_t pad[3];
} pm_photon;

typedef struct pm_ctx pm_ctx;   /* replaces the scheduler's global state, src/JobSystem.cpp:29-35 */
typedef struct pm_map pm_map;   /* replaces kdnode* (src/PhotonMap.h:15-20) */

#define PM_MAP_GLOBAL 0         /* createPhotonMap */
#define PM_MAP_CAUSTIC 1        /* createCausticPhotonMap */

#define PM_FLAG_NO_GLOBAL 1     /* skip the global-map gather (src/Raytrace.cpp:96-131) */
#define PM_FLAG_NO_CAUSTIC 2    /* skip the caustic-map gather (:135-171) */
#define PM_FLAG_NO_DIRECT 4     /* zero the shadow-ray direct term (:174-212) */
#define PM_FLAG_COUNT_TESTS 8   /* diagnostics: the primary and the shadow pass count the SphereObject / PlaneObject::intersect calls they
                                 * make (pm_stats.tests_*); same pane, slower kernels -- never the frame that is timed */

#define PM_K 63                 /* PHOTONS_IN_ESTIMATE, src/Raytrace.cpp:16,21 (heap only defined for 63, PhotonMap.cpp:174) */
#define PM_ROW_BLOCK 16         /* rows per interleaved row block when a frame is split over ranks */

typedef struct pm_render_params {
    int32_t width, height;      /* pane size (any W x H; the 16:9 rule lives in Main.cpp only) */
    double cam[3];              /* camera, src/Image.cpp:33 */
    double light_color[3];      /* src/Raytrace.cpp:311 */
    uint64_t seed;
    int32_t flags;              /* PM_FLAG_* */
    int32_t rank, world;        /* this call renders row blocks b with b % world == rank (rows 16b..16b+15) */
    int32_t reserved;
} pm_render_params;

typedef struct pm_stats {
    uint64_t closest_hit_queries;  /* Scene::intersect calls made by traceRay */
    uint64_t shadow_queries;       /* shadow any-hit queries (25 per absorb point) */
    uint64_t gathers;              /* find_nearest_photons top-level calls */
    uint64_t absorb_points;
    uint64_t complex_pixels;       /* pixels whose primary hit reflects or refracts */
    /* device time of each phase of the last call, from HIP events on the stream the phase runs on (ms).  The two maps are
     * gathered side by side on two streams: ms_gather_global and ms_gather_caustic overlap, ms_gather spans both. */
    float ms_primary, ms_secondary, ms_shadow, ms_gather_global, ms_gather_caustic, ms_resolve, ms_total;
    float ms_gather;
    /* the group kernel of each map's gather (one launch each) and the shade points it answered; the per-query kernel
     * answers the rest */
    float ms_group_global, ms_group_caustic;
    uint32_t grouped_global, grouped_caustic;
    /* with PM_FLAG_COUNT_TESTS: primitive intersection tests executed by the primary closest-hit pass and by the shadow any-hit pass
     * (a scene of 64 spheres or more is walked through the sphere grid: far fewer than rays x primitives) */
    uint64_t tests_primary_sphere, tests_primary_plane, tests_shadow_sphere, tests_shadow_plane;
    /* the per-query kernel's share: shade points the group kernel of each map handed over (gathered by the per-query launch
     * behind it, ms_gather_* - ms_group_*), and the bounced pixels' shade points that were never grouped (gathered by a
     * per-query launch beside the group kernels, once per map) */
    uint32_t left_global, left_caustic, early_points, reserved;
} pm_stats;

/* scheduler::startWorkers(int), src/JobSystem.cpp:76-86 (called at src/Main.cpp:34): `device` takes the place of `cores`.
 * The context creates three streams of its own.  In a process that also uses RCCL, create the context first: hardware
 * queues (4 per process) are dealt to streams in creation order, and streams that share one run one behind the other. */
int pm_ctx_create(int device, pm_ctx **out);
int pm_ctx_destroy(pm_ctx *ctx);
/* scheduler::waitForCompletion(), src/JobSystem.cpp:109-118: returns when everything the context has queued (the caller's
 * stream and the library's own three: two for builds and gathers, one for the ordering of bounced shade points) is done.  Unlike the reference's, the context stays usable. */
int pm_ctx_synchronize(pm_ctx *ctx);
/* Switches of a context that are no part of the reference's interface (it reads nothing at run time: its tunables are
 * #defines, src/Raytrace.cpp:11-27).  "debug" (0 / 1): per-frame statistics on stderr; starts as 1 when PM_DEBUG is in the
 * environment while the context is made -- the library reads the environment then and when a scene is uploaded, never per
 * frame.  "sp_cap_limit" (slots, 0 = off; tests): pretend the shade point queue is this small, to show that a frame that
 * overflows it is reported (PM_ESTATE), not delivered. */
int pm_ctx_set_option(pm_ctx *ctx, const char *name, double value);
/* Issue all work on this hipStream_t (NULL = the null stream). */
int pm_ctx_set_stream(pm_ctx *ctx, void *hip_stream);
const char *pm_last_error(void);

/* Scene + SceneObject*[] (src/Scene.h:24-34): flattened copy, index = object id.  With 64 spheres or more the library
 * also builds a uniform grid over the spheres (host side, here): Scene::intersect's linear scan (src/Scene.cpp:25-47)
 * then tests only the primitives along the ray and returns the very hit the scan would (PM_SCENE_GRID=0: never). */
int pm_scene_upload(pm_ctx *ctx, const pm_prim *prims, int32_t n);

/* The outer while-loop body of createPhotonMap / createCausticPhotonMap (src/PhotonMap.cpp:288-367,
 * 400-484) for emission attempts [attempt_begin, attempt_begin + attempt_count), one attempt per
 * lane, compacted in attempt order.  d_out is DEVICE memory for `cap` records; *out_count receives
 * the number of successful attempts in the range (may exceed cap; only the first cap are written). */
int pm_emit_range(pm_ctx *ctx, int32_t map_kind, uint64_t seed, const double light_color[3],
                  uint64_t attempt_begin, uint64_t attempt_count, pm_photon *d_out, uint64_t cap,
                  uint64_t *out_count);

/* pm_emit_range without any allocation.  No counterpart in the reference: it is for callers that cannot call hipMalloc
 * (tight serving loops, harnesses that forbid allocation).  It allocates nothing, but like pm_emit_range it WAITS for the
 * stream to hand back *out_count, so it cannot be part of a HIP graph capture.  Same arguments, same records and count
 * as pm_emit_range, bit for bit; the kernels' scratch (attempt records, flags, block counts and offsets, 4.4 MB) is a
 * static device array that every GPU the library runs on holds until the process ends, so attempt_count is limited to
 * PM_EMIT_STATIC_MAX_ATTEMPTS (more: PM_EINVAL).  Calls on one GPU share that array: calls that overlap on one GPU race. */
#define PM_EMIT_STATIC_MAX_ATTEMPTS 131072
int pm_emit_range_static(pm_ctx *ctx, int32_t map_kind, uint64_t seed, const double light_color[3],
                         uint64_t attempt_begin, uint64_t attempt_count, pm_photon *d_out, uint64_t cap,
                         uint64_t *out_count);

/* pm_emit_range for one rank of a multi-GPU job, without a wait: the shard is DEVICE memory of 32 + 32 * cap bytes, a
 * 32-byte header {uint64 count, uint64 cap, 0, 0} followed by the first cap successes of the range in attempt order.
 * count = successes in the range (it may exceed cap).  Everything is queued on the context's stream; the shards of all
 * ranks are then exchanged with ONE all-gather (RCCL) and handed to pm_map_build_shards.  No counterpart in the
 * reference (it has no ranks); the loop body is the one pm_emit_range cites. */
int pm_emit_shard(pm_ctx *ctx, int32_t map_kind, uint64_t seed, const double light_color[3],
                  uint64_t attempt_begin, uint64_t attempt_count, void *d_shard, uint64_t cap);

/* createKDTree (src/PhotonMap.cpp:64-140) on the first n records, in rank order, of `world` shards laid out
 * shard_stride bytes apart in DEVICE memory (the result of the all-gather).  Nothing waits: the first use of the map
 * does, and fails with PM_ESTATE if the shards held fewer than n photons or one overflowed its cap. */
int pm_map_build_shards(pm_ctx *ctx, const void *d_shards, int32_t world, uint64_t shard_stride, int64_t n, pm_map **out);

/* createKDTree (src/PhotonMap.cpp:64-140): builds the spatial index (linear octree over
 * Morton-sorted photons) from n records in emission order.  records_on_device: 0 = host pointer, 1 = device.
 * The records are copied in stream order (the caller may reuse them behind the call); the build is left on one of the
 * context's build streams and the first use of the map waits for it. */
int pm_map_build(pm_ctx *ctx, const pm_photon *records, int64_t n, int32_t records_on_device, pm_map **out);

/* createPhotonMap / createCausticPhotonMap (src/PhotonMap.h:28-29): the first n successful attempts,
 * in attempt order from attempt 0, then pm_map_build.  With attempts_used == NULL the call returns at once: emission
 * and build run on one of the context's two build streams (two maps side by side, beside the first passes of the frame
 * that needs them) and the first use of the map waits.  Otherwise the call waits and *attempts_used receives the
 * attempts issued. */
int pm_map_create(pm_ctx *ctx, int32_t map_kind, int64_t n, uint64_t seed, const double light_color[3],
                  pm_map **out, uint64_t *attempts_used);
int pm_map_size(const pm_map *map, int64_t *n);
/* The two phase times createPhotonMap prints (src/PhotonMap.cpp:369-371 "photons traced in", :378-382 "kd-tree built in"):
 * device time of the map's emission and of its index build, from HIP events on the stream they ran on (ms).  Waits for the map.
 * A map that had to be topped up synchronously reports the first, asynchronous part only. */
int pm_map_times(pm_map *map, float *ms_emit, float *ms_build);
/* Copies the n records, in emission order, to host memory. */
int pm_map_download(pm_map *map, pm_photon *host_out);
/* deleteTree, src/PhotonMap.cpp:504-513 */
int pm_map_destroy(pm_map *map);

/* find_nearest_photons (src/PhotonMap.cpp:210-255) with k = 63, max_dist = 100 for m query points
 * (host arrays; points m x 3 doubles).  Per query: found (<= 63), then idx / d2 ascending by
 * (d2, emission index); unused slots are -1 / 0. */
int pm_knn_query(pm_ctx *ctx, pm_map *map, int64_t m, const double *points, int32_t *idx, double *d2,
                 int32_t *found);

/* pm_knn_query plus per-query work counters (diagnostics; no counterpart in the reference):
 * work[4q..4q+3] = cell splits, chunks of 64 photons read, buffer reduces, candidates collected. */
int pm_knn_profile(pm_ctx *ctx, pm_map *map, int64_t m, const double *points, int32_t *idx, double *d2,
                   int32_t *found, uint32_t *work);

/* find_nearest_photons + the radiance loop of src/Raytrace.cpp:97-131 (map_kind GLOBAL) or :136-170
 * (CAUSTIC) for m points; radiance is m x 3 doubles.  on_device: 0 = host arrays, 1 = device arrays. */
int pm_knn_gather(pm_ctx *ctx, pm_map *map, int32_t map_kind, int64_t m, const double *points,
                  const double *normals, double *radiance, int32_t on_device);

/* render_task up to the first Scene::intersect (src/Raytrace.cpp:284-293,:39): hit ids (-1 = miss),
 * hit points and normals (NULL to skip) for rows y0 <= y < y1; host arrays, (y1-y0) x width entries. */
int pm_primary_hits(pm_ctx *ctx, const pm_render_params *rp, int32_t y0, int32_t y1, int32_t *ids,
                    double *points, double *normals);

/* SceneObject::intersect (src/Scene.h:11; SphereObject src/Scene.cpp:87-116, PlaneObject :135-193) and Scene::intersect
 * (src/Scene.h:32, src/Scene.cpp:25-47) for n caller-given rays, one lane per ray, through the device functions the frame
 * kernels use.  Host arrays: src, dir n x 3 doubles (dir as given: not normalised here), dt n doubles (NULL = 0).
 * prim[i] >= 0: that primitive alone; hit[i] = 1 / 0.  prim[i] < 0 or prim == NULL: closest hit over the scene, skipping
 * exclude[i] (NULL = nothing); hit[i] = object id or -1.  res / nrm (n x 3 doubles) are written on a hit and left as they
 * were on a miss, like the reference's out-parameters. */
int pm_intersect_rays(pm_ctx *ctx, int64_t n, const double *src, const double *dir, const double *dt,
                      const int32_t *prim, const int32_t *exclude, int32_t *hit, double *res, double *nrm);

/* The submit loop + waitForCompletion of renderScene (src/Raytrace.cpp:322-336): traces this rank's
 * row blocks into `pane` (full width x height image, pane[x + y*width] = 0xAARRGGBB; rows of other
 * ranks are left untouched).  pane_on_device: 0 = host, 1 = device.  gmap / cmap may be NULL only
 * with the matching PM_FLAG_NO_* flag.  stats may be NULL. */
int pm_render(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, uint32_t *pane,
              int32_t pane_on_device, pm_stats *stats);

/* The image front-end of raytrace::render (src/Image.cpp:36-57): out[x + y'*width] = per-channel integer mean (truncating
 * division) of the sample_ratio x sample_ratio block of `pane` (a (width*sample_ratio) x (height*sample_ratio) image),
 * y' = height-1-y when flip != 0 (the reference flips), alpha 0xFF, words left as 0xAARRGGBB.  Either side host or device. */
int pm_downsample(pm_ctx *ctx, const uint32_t *pane, int32_t pane_on_device, int32_t width, int32_t height,
                  int32_t sample_ratio, int32_t flip, uint32_t *out, int32_t out_on_device);

/* pm_render into a pane kept on the device + pm_downsample: rp->width x rp->height is the supersampled pane, `out` the
 * (rp->width / sample_ratio) x (rp->height / sample_ratio) image (src/Image.cpp:32-57 without the 4 s^2-byte-per-pixel
 * round trip through the host).  world must be 1. */
int pm_render_image(pm_ctx *ctx, pm_map *gmap, pm_map *cmap, const pm_render_params *rp, int32_t sample_ratio,
                    int32_t flip, uint32_t *out, int32_t out_on_device, pm_stats *stats);

#ifdef __cplusplus
}
#endif
#endif

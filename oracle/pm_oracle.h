/*
 * oracle/pm_oracle.h -- TEST INFRASTRUCTURE (oracle). Not part of the product.
 *
 * CPU restatement, in plain C, of the reference hot path (Deamon5550/raytracer):
 * src/Raytrace.cpp:32-299, src/PhotonMap.cpp:13-501, src/Scene.cpp:25-193,
 * src/Vector.cpp:8-132.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (raytracer_amd/) never does.
 *
 * Parity status: PINNED.  oracle/ref_harness.cpp compiles the unmodified
 * reference sources from /root/reference (recipe: oracle/Makefile, outputs in
 * oracle/_ref/) and checks this restatement against them bit for bit under
 * tape-replayed random draws; the resulting vectors are committed under
 * tests/golden/ (the reference holds no tests or golden vectors of its own,
 * SURVEY section 4).
 */
#ifndef PM_ORACLE_H
#define PM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flattened SceneObject / SphereObject / PlaneObject (src/Scene.h:8-59). */
typedef struct orc_prim {
    int32_t kind;              /* 0 = SphereObject, 1 = PlaneObject */
    int32_t pad0;
    double x, y, z;            /* SceneObject::x,y,z */
    double radius;             /* SphereObject::radius */
    double dx, dy, dz;         /* SphereObject::dx,dy,dz (linear motion) */
    double min_bound, max_bound; /* PlaneObject bounds */
    float red, green, blue;    /* byte/255.0f  (src/Scene.cpp:54-56) */
    float pad1;
    double absorb_chance, diffuse_chance, specular_chance, transmission_chance;
    double refraction, specular_coeff;
} orc_prim;                    /* 144 bytes */

/* struct photon (src/PhotonMap.h:8-13), 32 bytes. */
typedef struct orc_photon {
    float x, y, z;
    uint8_t power[4];
    float dx, dy, dz;
    int8_t bounce;
    uint8_t pad[3];
} orc_photon;

typedef struct orc_tape { double *v; size_t n, cap; } orc_tape;

typedef struct orc_map orc_map;

enum { ORC_MAP_GLOBAL = 0, ORC_MAP_CAUSTIC = 1 };
enum { ORC_TRIG_LIBM = 0, ORC_TRIG_DET = 1 };
enum { ORC_ORDER_CANONICAL = 0, ORC_ORDER_REFERENCE = 1 };
enum { ORC_FLAG_NO_GLOBAL = 1, ORC_FLAG_NO_CAUSTIC = 2, ORC_FLAG_NO_DIRECT = 4 };

typedef struct orc_stats {
    uint64_t closest_hit_queries;  /* Scene::intersect calls from traceRay */
    uint64_t shadow_queries;       /* shadow any-hit queries */
    uint64_t gathers;              /* find_nearest_photons top-level calls */
    uint64_t absorb_points;
} orc_stats;

/* ---- primitives ---- */
void orc_default_scene(orc_prim out[7]);                    /* src/Image.cpp:20-28 */
int32_t orc_fastfloor(double x);                             /* src/Vector.cpp:13-16 */
int orc_prim_intersect(const orc_prim *p, const double src[3], const double dir[3], double dt,
                       double res[3], double nrm[3]);        /* src/Scene.cpp:87-116,135-193 */
int orc_scene_intersect(const orc_prim *prims, int n, const double src[3], const double dir[3],
                        int exclude, double dt, double res[3], double nrm[3]); /* src/Scene.cpp:25-47 */
void orc_det_sincos(double x, double *s, double *c);

/* ---- primary rays (src/Raytrace.cpp:284-293) ---- */
void orc_primary_ray(int width, int height, const double cam[3], uint64_t seed, int x, int y,
                     double dir[3]);
void orc_primary_hits(const orc_prim *prims, int n, int width, int height, const double cam[3],
                      uint64_t seed, int y0, int y1, int32_t *ids, double *hit, double *nrm);

/* ---- photon emission (src/PhotonMap.cpp:278-368, 391-485) ---- */
/* One emission attempt; returns 1 and fills *out if a photon is stored.  ndraws = draws consumed. */
int orc_emit_attempt(const orc_prim *prims, int n, int map_kind, uint64_t seed, uint64_t attempt,
                     int trig, const double light_color[3], orc_photon *out, int *ndraws,
                     orc_tape *tape);
/* First n_photons successes in attempt order starting at attempt 0.  Returns attempts used. */
uint64_t orc_emit(const orc_prim *prims, int n, int map_kind, uint64_t seed, int n_photons,
                  int trig, const double light_color[3], orc_photon *out, orc_tape *tape);

/* ---- photon maps ---- */
/* build_ref != 0 also builds the reference's own kd-tree shape (src/PhotonMap.cpp:13-140)
 * with pivots from stream kind 4 (pivot_seed) -- only sensible for n <= ~20k. */
orc_map *orc_map_create(const orc_photon *photons, int n, int build_ref, uint64_t pivot_seed);
void orc_map_destroy(orc_map *m);
int orc_map_size(const orc_map *m);
const orc_photon *orc_map_photons(const orc_map *m);
int orc_map_ref_depth(const orc_map *m);

/* find_nearest_photons (src/PhotonMap.cpp:210-255) on the reference-shaped tree:
 * heap arrays in the reference's order. Returns found. idx = photon index (emission order). */
int orc_knn_ref(const orc_map *m, const double q[3], int32_t idx[63], double d2[63]);
/* insert (src/PhotonMap.cpp:142-206) on caller arrays, `next` a photon index. */
void orc_heap_insert(int32_t *nearest, double *distances, int k, int size, int32_t next, double dist);
/* Exact k-NN (k=63, d2<100), ascending by (d2, idx). Returns found. */
int orc_knn_exact(const orc_map *m, const double q[3], int32_t idx[63], double d2[63]);
/* Brute force, same ordering as orc_knn_exact. */
int orc_knn_brute(const orc_map *m, const double q[3], int32_t idx[63], double d2[63]);

/* Radiance estimate (src/Raytrace.cpp:97-131 global, :136-170 caustic). */
void orc_gather(const orc_map *m, int map_kind, int order, const double p[3], const double nrm[3],
                double rgb[3]);
void orc_gather_batch(const orc_map *m, int map_kind, int order, int count, const double *p,
                      const double *nrm, double *rgb, int nthreads);

/* ---- tracing (src/Raytrace.cpp:32-299) ---- */
typedef struct orc_render_params {
    int32_t width, height;
    double cam[3];
    double light_color[3];
    uint64_t seed;
    int32_t flags;
    int32_t order;
} orc_render_params;

uint32_t orc_trace_pixel(const orc_prim *prims, int n, const orc_map *gmap, const orc_map *cmap,
                         const orc_render_params *rp, int x, int y, orc_tape *tape,
                         orc_stats *stats);
/* Rows y0 <= y < y1 with (y - y0) % ystep == 0, one job per row on nthreads workers
 * (src/Raytrace.cpp:322-336, src/JobSystem.cpp:37-73). pane is the full width*height image. */
void orc_render_rows(const orc_prim *prims, int n, const orc_map *gmap, const orc_map *cmap,
                     const orc_render_params *rp, int y0, int y1, int ystep, int nthreads,
                     uint32_t *pane, orc_stats *stats);

void orc_tape_init(orc_tape *t);
void orc_tape_free(orc_tape *t);
uint64_t orc_pivot_next(uint64_t pivot_seed, uint64_t *counter, int32_t lo, int32_t hi);

#ifdef __cplusplus
}
#endif
#endif

// raytrace_api.h -- the host-side C++ API a user of Deamon5550/raytracer already writes against,
// re-implemented above the C-ABI of libraytracer_amd.so (include/raytracer_amd.h).
//
// Same names, argument order and meaning as the reference for everything on the hot path:
//   raytrace::Vec3, Scene, SceneObject, SphereObject, PlaneObject        (reference src/Vector.h, src/Scene.h)
//   raytrace::photon, kdnode, createPhotonMap, createCausticPhotonMap,
//             find_nearest_photons, deleteTree                            (src/PhotonMap.h:8-31)
//   raytrace::renderScene                                                 (src/Raytrace.h:10)
//   raytrace::render                                                      (src/Image.h:8)
//   scheduler::startWorkers / submit / waitForCompletion                  (src/JobSystem.h:17-19)
//   randutil::init                                                        (src/Random.h:7)
// What differs, by design: the CPU worker pool is gone (startWorkers opens the GPU context, jobs are
// kernels), kdnode is an opaque handle to a GPU photon map, and random numbers come from counter-based
// streams keyed by the seed given to randutil::init (RNG contract v1).  Like the reference,
// these functions return void / pointers; an unrecoverable backend error prints a message and aborts.
#pragma once
#include <stdint.h>

typedef char int8;
typedef unsigned char uint8;
typedef int int32;
typedef unsigned int uint32;
typedef long long int64;
typedef unsigned long long uint64;

struct pm_map;
struct pm_prim;

namespace raytrace {

class Vec3 {
public:
    Vec3(double x0, double y0, double z0) : x(x0), y(y0), z(z0) {}
    explicit Vec3(Vec3 *s) : x(s->x), y(s->y), z(s->z) {}
    void set(double x0, double y0, double z0) { x = x0; y = y0; z = z0; }
    void set(Vec3 *o) { x = o->x; y = o->y; z = o->z; }
    double x, y, z;
};

class SceneObject {
public:
    virtual ~SceneObject() {}
    // fills the flattened record the GPU uses; index in Scene::objects is the hit id
    virtual void flatten(pm_prim *out) const = 0;
    double x, y, z;
    float red, green, blue;
    double absorb_chance, diffuse_chance, specular_chance, transmission_chance, refraction, specular_coeff;
protected:
    void material(uint32 col, double d, double s, double t, double a);
    void flatten_common(pm_prim *out) const;
};

class Scene {
public:
    explicit Scene(int32 num_objects);
    ~Scene();
    int32 size;
    SceneObject **objects;
};

class SphereObject : public SceneObject {
public:
    SphereObject(double x0, double y0, double z0, double r0, uint32 col, double d, double s, double t, double a);
    SphereObject(double x0, double y0, double z0, double r0, uint32 col, double d, double s, double t, double a, double dx, double dy, double dz);
    void flatten(pm_prim *out) const override;
    double radius;
    double dx, dy, dz;
};

class PlaneObject : public SceneObject {
public:
    PlaneObject(double x0, double y0, double z0, double min, double max, uint32 col, double d, double s, double t, double a);
    void flatten(pm_prim *out) const override;
    double min_bound, max_bound;
};

struct photon {
    float x, y, z;
    uint8 power[4];
    float dx, dy, dz;
    int8 bounce;
};

// Opaque: a photon map resident on the GPU (the reference's pointer tree has no counterpart there).
struct kdnode {
    pm_map *map;
    int32 kind;        // 0 global, 1 caustic
    int32 count;
    photon *records;   // host copy in emission order, fetched on first use by find_nearest_photons
};

kdnode *createPhotonMap(int32 photon_count, Vec3 &light_source, Vec3 &light_color, Scene *scene);
kdnode *createCausticPhotonMap(int32 photon_count, Vec3 &light_source, Vec3 &light_color, Scene *scene);
// k must be 63 and size 0 (the reference's heap is only defined for them); fills nearest / distances,
// ascending by distance, and returns the number found.
int find_nearest_photons(photon **nearest, double *distances, int k, int size, Vec3 *target, kdnode *root, double max_dist);
void deleteTree(kdnode *tree);

void renderScene(Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height);
void render(const char *image_file, int32 width, int32 height, int32 sample_ratio);

// number of photons renderScene asks for per map (reference: compile-time 2048, src/Raytrace.cpp:11,19)
void setPhotonCounts(int32 global_photons, int32 caustic_photons);

}  // namespace raytrace

namespace scheduler {
typedef void (*task)(void *);
void startWorkers(int count);              // opens the GPU context (count is accepted and ignored)
void submit(task job, void *job_data);     // runs the job on the calling thread (kept for source compatibility)
void waitForCompletion();                  // waits for the GPU stream
}

namespace randutil {
void init(int64 seed);                     // seed of the counter-based streams used from now on
}

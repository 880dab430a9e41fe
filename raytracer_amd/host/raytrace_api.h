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

// reference src/Vector.h:14-16, src/Vector.cpp:8-20 (host arithmetic: no ray is traced with these)
int32 fastfloor(float value);
int32 fastfloor(double value);
int32 min(int32 a, int32 b);

// reference src/Vector.h:19-45: three doubles, operations in place.  Host-side value type (the kernels keep their own V3
// with the same operation order, pm_device.h); normalize() is three DIVISIONS by the length, as at src/Vector.cpp:95-100.
class Vec3 {
public:
    Vec3(double x0, double y0, double z0) : x(x0), y(y0), z(z0) {}
    explicit Vec3(Vec3 *s) : x(s->x), y(s->y), z(s->z) {}
    void set(double x0, double y0, double z0) { x = x0; y = y0; z = z0; }
    void set(Vec3 *o) { x = o->x; y = o->y; z = o->z; }
    void add(double x0, double y0, double z0) { x += x0; y += y0; z += z0; }
    void add(Vec3 *o) { add(o->x, o->y, o->z); }
    void sub(double x0, double y0, double z0) { x -= x0; y -= y0; z -= z0; }
    void sub(Vec3 *o) { sub(o->x, o->y, o->z); }
    void mul(double k) { x *= k; y *= k; z *= k; }
    void mul(double x0, double y0, double z0) { x *= x0; y *= y0; z *= z0; }
    double lengthSquared() { return x * x + y * y + z * z; }
    double length();
    void normalize() { const double len = length(); x /= len; y /= len; z /= len; }
    double dot(Vec3 *o) { return dot(o->x, o->y, o->z); }
    double dot(double ox, double oy, double oz) { return x * ox + y * oy + z * oz; }
    double distSquared(Vec3 *o) { return (x - o->x) * (x - o->x) + (y - o->y) * (y - o->y) + (z - o->z) * (z - o->z); }
    double distSquared(float ox, float oy, float oz) { return (x - ox) * (x - ox) + (y - oy) * (y - oy) + (z - oz) * (z - oz); }
    double x, y, z;
};
Vec3 cross(Vec3 *a, Vec3 *b);
enum Axis { X_AXIS, Y_AXIS, Z_AXIS };
Axis largestAxis(double x, double y, double z);

class SceneObject {
public:
    virtual ~SceneObject() {}
    // reference src/Scene.h:11.  One ray on the GPU (pm_intersect_rays): the object is tested where it stands in the scene
    // that was uploaded last, or on its own when it belongs to none.  true: *result / *result_normal hold the hit.
    virtual bool intersect(Vec3 *ray_source, Vec3 *ray, Vec3 *result, Vec3 *result_normal, double dt);
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
    // reference src/Scene.h:32, src/Scene.cpp:25-47: closest hit, `exclude` skipped, *hit_object = nullptr on a miss (then
    // *result / *result_normal are left alone).  One ray on the GPU (pm_intersect_rays); whole frames go through renderScene.
    void intersect(Vec3 &ray_source, Vec3 &ray, SceneObject *exclude, Vec3 *result, Vec3 *result_normal, SceneObject **hit_object, double dt);
};
// The Cornell box of src/Image.cpp:20-28 (what raytrace::render builds when it is given no scene).
Scene *defaultScene();
// A scene description file.  Binary (`*.bin`): an array of pm_prim records (include/raytracer_amd.h; tests/golden/scene_*.bin).
// Text: one object per line, arguments in the order of the reference's constructors, colour as hex AARRGGBB or RRGGBB:
//   plane  x y z  min max  colour  d s t a  [refraction]
//   sphere x y z  radius   colour  d s t a  [refraction [dx dy dz]]
// '#' starts a comment.  Returns nullptr (and prints why) when the file cannot be read.
Scene *loadScene(const char *file);
// The synthetic scenes of BASELINE configs 4 and 5 (no counterpart in the reference, whose only scene is the Cornell box):
// five glass spheres; the five walls + n small diffuse spheres from a counter-based stream (same scene as
// raytracer_amd/scenes.py, byte for byte).
Scene *causticsScene();
Scene *manySpheresScene(int32 n_spheres, uint64 seed);

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
// reference src/PhotonMap.h:24, src/PhotonMap.cpp:142-206: one step of the 63-entry max-heap on caller arrays (host
// arithmetic; the GPU search keeps no heap).  Like the reference's, the sift-down is only defined for k = 63.
void insert(photon **nearest, double *distances, int k, int size, photon *next, double dist);
void deleteTree(kdnode *tree);

void renderScene(Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height);
void render(const char *image_file, int32 width, int32 height, int32 sample_ratio);
// the same with a scene of the caller's (deleted by the call, as render deletes its own); nullptr = the Cornell box
void render(const char *image_file, int32 width, int32 height, int32 sample_ratio, Scene *scene);

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

// raytrace_api.cpp -- host C++ API of the reference (see raytrace_api.h) above the C-ABI.
// No ray is traced and no photon is stored on the CPU here: every function forwards to
// libraytracer_amd.so and aborts with a message if the GPU backend reports an error.
#include "raytrace_api.h"
#include "raytrace_backend.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/raytracer_amd.h"

namespace {
pm_ctx *g_ctx = nullptr;
uint64_t g_seed = 0;
bool g_seeded = false;
int32 g_global_photons = 2048;    // NUM_PHOTONS, reference src/Raytrace.cpp:11
int32 g_caustic_photons = 2048;   // CAUSTIC_PHOTONS, reference src/Raytrace.cpp:19
std::vector<pm_prim> g_uploaded;   // what the GPU holds: the scene is sent again only when its flattened records change

void check(int rc, const char *what) {
    if (rc != PM_OK) {
        std::fprintf(stderr, "raytracer_amd: %s failed (%d): %s\n", what, rc, pm_last_error());
        std::abort();
    }
}
pm_ctx *ctx() {
    if (!g_ctx) check(pm_ctx_create(0, &g_ctx), "pm_ctx_create");
    return g_ctx;
}
void upload(const std::vector<pm_prim> &prims) {
    if (prims.size() == g_uploaded.size() && (prims.empty() || std::memcmp(prims.data(), g_uploaded.data(), prims.size() * sizeof(pm_prim)) == 0)) return;
    check(pm_scene_upload(ctx(), prims.data(), (int32_t)prims.size()), "pm_scene_upload");
    g_uploaded = prims;
}
// the caller may have changed an object since the last call (the reference's own scene set-up assigns `refraction` after
// the constructor, src/Image.cpp:27): flatten every time, upload when something differs
void upload(raytrace::Scene *scene) {
    std::vector<pm_prim> prims((size_t)scene->size);
    for (int i = 0; i < scene->size; i++) {
        std::memset(&prims[(size_t)i], 0, sizeof(pm_prim));
        if (!scene->objects[i]) { std::fprintf(stderr, "raytracer_amd: scene object %d is null\n", i); std::abort(); }
        scene->objects[i]->flatten(&prims[(size_t)i]);
    }
    upload(prims);
}
uint64_t seed() {
    if (!g_seeded) {   // the reference seeds from the clock in raytrace::render (src/Image.cpp:16-17)
        g_seed = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        g_seeded = true;
    }
    return g_seed;
}
}  // namespace

// what raytrace_dist.cpp (the one-process-per-GPU frame) needs of this file's state
namespace raytrace { namespace backend {
pm_ctx *context() { return ctx(); }
void uploadScene(Scene *scene) { upload(scene); }
unsigned long long frameSeed() { return seed(); }
void photonCounts(int32 *global_photons, int32 *caustic_photons) { *global_photons = g_global_photons; *caustic_photons = g_caustic_photons; }
void checked(int rc, const char *what) { check(rc, what); }
bool writeImage(const char *file, int width, int height, const uint32 *words);
} }

namespace raytrace {

int32 fastfloor(float x) { const int32 xi = (int32)x; return x < xi ? xi - 1 : xi; }
int32 fastfloor(double x) { const int32 xi = (int32)x; return x < xi ? xi - 1 : xi; }
int32 min(int32 a, int32 b) { return a < b ? a : b; }
double Vec3::length() { return std::sqrt(x * x + y * y + z * z); }
Vec3 cross(Vec3 *a, Vec3 *b) { return Vec3(a->y * b->z - a->z * b->y, a->z * b->x - a->x * b->z, a->x * b->y - a->y * b->x); }
Axis largestAxis(double x, double y, double z) { return x > y ? (x > z ? X_AXIS : Z_AXIS) : (y > z ? Y_AXIS : Z_AXIS); }

namespace {
// one ray through pm_intersect_rays; prim < 0: closest hit over the uploaded scene
int one_ray(Vec3 *src, Vec3 *ray, double dt, int32_t prim, int32_t exclude, Vec3 *result, Vec3 *normal) {
    const double o[3] = {src->x, src->y, src->z}, d[3] = {ray->x, ray->y, ray->z};
    double r[3] = {result->x, result->y, result->z}, n[3] = {normal->x, normal->y, normal->z};
    int32_t hit = 0;
    check(pm_intersect_rays(ctx(), 1, o, d, &dt, &prim, &exclude, &hit, r, n), "pm_intersect_rays");
    result->set(r[0], r[1], r[2]);      // left as they were on a miss: the library hands back what it was given
    normal->set(n[0], n[1], n[2]);
    return hit;
}
}  // namespace

bool SceneObject::intersect(Vec3 *ray_source, Vec3 *ray, Vec3 *result, Vec3 *result_normal, double dt) {
    // where does this object stand in the scene the GPU holds?  (its flattened record is the identity: two objects with
    // the same record intersect alike)
    pm_prim me;
    std::memset(&me, 0, sizeof(me));
    flatten(&me);
    int32_t at = -1;
    for (size_t i = 0; i < g_uploaded.size() && at < 0; i++)
        if (std::memcmp(&g_uploaded[i], &me, sizeof(me)) == 0) at = (int32_t)i;
    if (at >= 0) return one_ray(ray_source, ray, dt, at, -1, result, result_normal) != 0;
    // an object on its own: a scene of one for the length of this call, then the scene the GPU held before (with its sphere
    // grid and what the library has learnt about it) is put back
    const std::vector<pm_prim> before = g_uploaded;
    upload(std::vector<pm_prim>(1, me));
    const bool hit = one_ray(ray_source, ray, dt, 0, -1, result, result_normal) != 0;
    if (!before.empty()) upload(before);
    return hit;
}

void Scene::intersect(Vec3 &ray_source, Vec3 &ray, SceneObject *exclude, Vec3 *result, Vec3 *result_normal, SceneObject **hit_object, double dt) {
    upload(this);
    int32_t skip = -1;
    for (int i = 0; i < size && skip < 0; i++) if (objects[i] == exclude) skip = i;
    const int hit = one_ray(&ray_source, &ray, dt, -1, skip, result, result_normal);
    *hit_object = hit >= 0 ? objects[hit] : nullptr;
}

void SceneObject::material(uint32 col, double d, double s, double t, double a) {
    red = ((col >> 16) & 0xFF) / 255.0f;
    green = ((col >> 8) & 0xFF) / 255.0f;
    blue = (col & 0xFF) / 255.0f;
    absorb_chance = a; diffuse_chance = d; specular_chance = s; transmission_chance = t;
    refraction = 0; specular_coeff = 0;
}
void SceneObject::flatten_common(pm_prim *o) const {
    o->x = x; o->y = y; o->z = z;
    o->red = red; o->green = green; o->blue = blue;
    o->absorb_chance = absorb_chance; o->diffuse_chance = diffuse_chance; o->specular_chance = specular_chance;
    o->transmission_chance = transmission_chance; o->refraction = refraction; o->specular_coeff = specular_coeff;
}

Scene::Scene(int32 num_objects) : size(num_objects), objects(new SceneObject *[(size_t)num_objects]) {
    for (int i = 0; i < size; i++) objects[i] = nullptr;
}
Scene::~Scene() {
    for (int i = 0; i < size; i++) delete objects[i];
    delete[] objects;
}

SphereObject::SphereObject(double x0, double y0, double z0, double r0, uint32 col, double d, double s, double t, double a)
    : SphereObject(x0, y0, z0, r0, col, d, s, t, a, 0, 0, 0) {}
SphereObject::SphereObject(double x0, double y0, double z0, double r0, uint32 col, double d, double s, double t, double a, double dx0, double dy0, double dz0) {
    x = x0; y = y0; z = z0; radius = r0; dx = dx0; dy = dy0; dz = dz0;
    material(col, d, s, t, a);
}
void SphereObject::flatten(pm_prim *o) const {
    flatten_common(o);
    o->kind = 0; o->radius = radius; o->dx = dx; o->dy = dy; o->dz = dz;
}
PlaneObject::PlaneObject(double x0, double y0, double z0, double min, double max, uint32 col, double d, double s, double t, double a) {
    x = x0; y = y0; z = z0; min_bound = min; max_bound = max;
    material(col, d, s, t, a);
}
void PlaneObject::flatten(pm_prim *o) const {
    flatten_common(o);
    o->kind = 1; o->min_bound = min_bound; o->max_bound = max_bound;
}

static const char *kKind[2] = {"Global", "Caustic"};
// the two lines the reference prints per map (src/PhotonMap.cpp:369-371, 378-382), with the device times of the two phases
static void print_map_times(kdnode *node) {
    float ms_emit = 0, ms_build = 0;
    check(pm_map_times(node->map, &ms_emit, &ms_build), "pm_map_times");
    std::printf("%s photons traced in %.3fs\n", kKind[node->kind], ms_emit * 1e-3);
    std::printf("%s photons kd-tree built in %.3fs\n", kKind[node->kind], ms_build * 1e-3);
}
// wait = false: emission and build are left running on the GPU (renderScene starts both maps, then the frame)
static kdnode *make_map(int kind, int32 n, Vec3 &light_source, Vec3 &light_color, Scene *scene, bool wait) {
    std::printf("Building %s photon map from %d photons\n", kind == PM_MAP_GLOBAL ? "global" : "caustic", n);
    upload(scene);
    double lc[3] = {light_color.x, light_color.y, light_color.z};
    kdnode *node = new kdnode{nullptr, kind, n, nullptr};
    check(pm_map_create(ctx(), kind, n, seed(), lc, &node->map, nullptr), "pm_map_create");
    // the reference leaves the last emission point in light_source (src/PhotonMap.cpp:293); nothing reads it
    (void)light_source;
    if (wait) print_map_times(node);
    return node;
}
kdnode *createPhotonMap(int32 n, Vec3 &light_source, Vec3 &light_color, Scene *scene) { return make_map(PM_MAP_GLOBAL, n, light_source, light_color, scene, true); }
kdnode *createCausticPhotonMap(int32 n, Vec3 &light_source, Vec3 &light_color, Scene *scene) { return make_map(PM_MAP_CAUSTIC, n, light_source, light_color, scene, true); }

// reference src/PhotonMap.cpp:142-206, restated: a max-heap on `distances` with `nearest` carried along
namespace {
inline void heap_swap(photon **nearest, double *distances, int a, int b) {
    const double d = distances[a]; distances[a] = distances[b]; distances[b] = d;
    photon *p = nearest[a]; nearest[a] = nearest[b]; nearest[b] = p;
}
}  // namespace
void insert(photon **nearest, double *distances, int k, int size, photon *next, double dist) {
    if (size < k) {                       // room left: append, sift up while the parent is smaller
        int i = size;
        nearest[i] = next; distances[i] = dist;
        for (int parent; i != 0 && distances[parent = (i - 1) / 2] < distances[i]; i = parent) heap_swap(nearest, distances, parent, i);
        return;
    }
    if (dist > distances[0]) return;      // full: only a photon no farther than the root gets in, replacing it
    nearest[0] = next; distances[0] = dist;
    for (int i = 0; i < 31;) {            // 31 inner nodes: k = 63 (the reference's literal)
        const int l = 2 * i + 1, r = l + 1;
        const int child = distances[l] < distances[r] ? r : l;   // the larger child, the left one on a tie
        if (!(distances[child] > distances[i])) break;
        heap_swap(nearest, distances, i, child);
        i = child;
    }
}

int find_nearest_photons(photon **nearest, double *distances, int k, int size, Vec3 *target, kdnode *root, double max_dist) {
    // k = 63 is all the reference's own insert() can do (its sift-down is written out for 31 inner nodes, src/PhotonMap.cpp:142-206),
    // and a search starts from an empty heap (size = 0).  max_dist is the reference's `dist < max_dist` (:215): any value up to
    // the 100 the device search is compiled for is served by cutting the ascending result short.
    if (k != PM_K || size != 0 || !(max_dist > 0) || max_dist > 100) {
        std::fprintf(stderr, "raytracer_amd: find_nearest_photons supports k = 63, size = 0, 0 < max_dist <= 100 (the reference calls it with 63, 0, 100)\n");
        std::abort();
    }
    if (!root->records && root->count > 0) {
        static_assert(sizeof(photon) == sizeof(pm_photon), "photon layout");
        root->records = new photon[(size_t)root->count];
        check(pm_map_download(root->map, reinterpret_cast<pm_photon *>(root->records)), "pm_map_download");
    }
    double q[3] = {target->x, target->y, target->z};
    int32_t idx[PM_K], found = 0;
    check(pm_knn_query(ctx(), root->map, 1, q, idx, distances, &found), "pm_knn_query");
    while (found > 0 && !(distances[found - 1] < max_dist)) found--;   // (ascending: what is not below max_dist is a suffix)
    for (int i = 0; i < PM_K; i++) nearest[i] = i < found ? &root->records[idx[i]] : nullptr;
    return found;
}
void deleteTree(kdnode *tree) {
    if (!tree) return;
    check(pm_map_destroy(tree->map), "pm_map_destroy");
    delete[] tree->records;
    delete tree;
}

void setPhotonCounts(int32 g, int32 c) { g_global_photons = g; g_caustic_photons = c; }

namespace {
// renderScene's body; image != nullptr: the pane stays on the device and `image` receives the box-filtered, flipped
// (width / s) x (height / s) picture (raytrace::render); else `pane` receives the width x height pane
void render_frame(Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height, int32 s, uint32 *image) {
    Vec3 light_source(0, 4.96, 4);
    Vec3 light_color(0.6, 0.6, 0.6);
    // both maps are started before either is waited for: they are emitted and indexed side by side, beside the first
    // passes of the frame.  Their phase times are therefore printed after the frame, not before it as in the reference.
    kdnode *global_tree = make_map(PM_MAP_GLOBAL, g_global_photons, light_source, light_color, scene, false);
    kdnode *caustic_tree = make_map(PM_MAP_CAUSTIC, g_caustic_photons, light_source, light_color, scene, false);
    std::printf("Rendering scene\n");
    auto t0 = std::chrono::high_resolution_clock::now();
    pm_render_params rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.width = width; rp.height = height;
    rp.cam[0] = camera.x; rp.cam[1] = camera.y; rp.cam[2] = camera.z;
    rp.light_color[0] = light_color.x; rp.light_color[1] = light_color.y; rp.light_color[2] = light_color.z;
    rp.seed = seed(); rp.rank = 0; rp.world = 1;
    pm_stats st;
    if (image) check(pm_render_image(ctx(), global_tree->map, caustic_tree->map, &rp, s, 1, image, 0, &st), "pm_render_image");
    else check(pm_render(ctx(), global_tree->map, caustic_tree->map, &rp, pane, 0, &st), "pm_render");
    std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
    print_map_times(global_tree);
    print_map_times(caustic_tree);
    std::printf("Scene rendered in %.3fs\n", dt.count());
    deleteTree(global_tree);
    deleteTree(caustic_tree);
}
}  // namespace

void renderScene(Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height) { render_frame(scene, camera, pane, width, height, 1, nullptr); }

// ---- image front-end: box downsample, flip, PNG (reference src/Image.cpp:14-76) -----------------
namespace {
uint32_t crc_table[256];
void crc_init() {
    for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
}
uint32_t crc(const unsigned char *p, size_t n, uint32_t c = 0xFFFFFFFFu) { for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8); return c; }
void be32(std::vector<unsigned char> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); }
void chunk(std::vector<unsigned char> &png, const char *type, const std::vector<unsigned char> &data) {
    be32(png, (uint32_t)data.size());
    size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    be32(png, crc(&png[at], png.size() - at) ^ 0xFFFFFFFFu);
}
// 8-bit RGBA PNG with stored (uncompressed) deflate blocks
bool write_png(const char *file, int w, int h, const unsigned char *rgba) {
    crc_init();
    std::vector<unsigned char> raw;
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4); }
    std::vector<unsigned char> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (unsigned char c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        size_t len = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + len >= raw.size() ? 1 : 0);
        z.push_back((unsigned char)(len & 0xFF)); z.push_back((unsigned char)(len >> 8));
        z.push_back((unsigned char)(~len & 0xFF)); z.push_back((unsigned char)((~len >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + len));
        if (raw.empty()) break;
    }
    be32(z, (b << 16) | a);
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr); chunk(png, "IDAT", z); chunk(png, "IEND", {});
    FILE *f = std::fopen(file, "wb");
    if (!f) return false;
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    return ok;
}
}  // namespace

Scene *defaultScene() {
    Scene *scene = new Scene(7);   // the Cornell box of src/Image.cpp:20-28
    scene->objects[0] = new PlaneObject(0, -5, 0, -5, 5, 0xFFEEEEEE, 0.4, 0.0, 0.0, 1.0);
    scene->objects[1] = new PlaneObject(0, 5, 0, -5, 5, 0xFDEEEEEE, 0.4, 0.0, 0.0, 1.0);   // the reference's 9-digit literal truncates to this
    scene->objects[2] = new PlaneObject(5, 0, 0, -5, 5, 0xFFFF3333, 0.4, 0.0, 0.0, 1.0);
    scene->objects[3] = new PlaneObject(-5, 0, 0, -5, 5, 0xFF3333FF, 0.4, 0.0, 0.0, 1.0);
    scene->objects[4] = new PlaneObject(0, 0, 10, -5, 5, 0xFFEEEEEE, 0.4, 0.0, 0.0, 1.0);
    scene->objects[5] = new SphereObject(2, -3.5, 3, 1.5, 0xFFFFFFFF, 0.0, 0.1, 0.9, 0.0);
    scene->objects[5]->refraction = 2.5;
    scene->objects[6] = new SphereObject(-2, -3.5, 5, 1.5, 0xFFFFFFFF, 0.0, 1.0, 0.0, 0.0);
    return scene;
}

// ---- scenes beyond the reference's one ------------------------------------
namespace {
// a scene object straight from a flattened record (scene files; the synthetic scenes)
class RecordSphere : public SphereObject {
public:
    explicit RecordSphere(const pm_prim &p) : SphereObject(p.x, p.y, p.z, p.radius, 0, p.diffuse_chance, p.specular_chance, p.transmission_chance, p.absorb_chance, p.dx, p.dy, p.dz) {
        red = p.red; green = p.green; blue = p.blue; refraction = p.refraction; specular_coeff = p.specular_coeff;
    }
};
class RecordPlane : public PlaneObject {
public:
    explicit RecordPlane(const pm_prim &p) : PlaneObject(p.x, p.y, p.z, p.min_bound, p.max_bound, 0, p.diffuse_chance, p.specular_chance, p.transmission_chance, p.absorb_chance) {
        red = p.red; green = p.green; blue = p.blue; refraction = p.refraction; specular_coeff = p.specular_coeff;
    }
};
Scene *scene_from_records(const std::vector<pm_prim> &prims) {
    Scene *scene = new Scene((int32)prims.size());
    for (size_t i = 0; i < prims.size(); i++)
        scene->objects[i] = prims[i].kind == 0 ? static_cast<SceneObject *>(new RecordSphere(prims[i])) : static_cast<SceneObject *>(new RecordPlane(prims[i]));
    return scene;
}
uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
bool ends_with(const std::string &s, const char *tail) { const size_t n = std::strlen(tail); return s.size() >= n && s.compare(s.size() - n, n, tail) == 0; }
}  // namespace

Scene *causticsScene() {
    static const double glass[5][4] = {{2.0, -3.5, 3.0, 1.5}, {-2.0, -3.5, 5.0, 1.5}, {0.0, -4.0, 7.5, 1.0}, {-3.0, -1.0, 8.0, 0.8}, {2.5, 0.5, 6.5, 1.0}};
    Scene *walls = defaultScene();
    Scene *scene = new Scene(10);
    for (int i = 0; i < 5; i++) { scene->objects[i] = walls->objects[i]; walls->objects[i] = nullptr; }
    delete walls;
    for (int i = 0; i < 5; i++) {   // the material of the reference's glass sphere, src/Image.cpp:26-27
        scene->objects[5 + i] = new SphereObject(glass[i][0], glass[i][1], glass[i][2], glass[i][3], 0xFFFFFFFF, 0.0, 0.1, 0.9, 0.0);
        scene->objects[5 + i]->refraction = 2.5;
    }
    return scene;
}

Scene *manySpheresScene(int32 n_spheres, uint64 seed) {
    // u_j = (mix64(mix64(seed) + (j + 1) * GOLD) >> 11) * 2^-53, seven per sphere: x y z radius red green blue
    const uint64_t base = mix64((uint64_t)seed), gold = 0x9E3779B97F4A7C15ULL;
    auto u = [&](uint64_t j) { return (double)(mix64(base + (j + 1) * gold) >> 11) * (1.0 / 9007199254740992.0); };
    Scene *walls = defaultScene();
    Scene *scene = new Scene(5 + n_spheres);
    for (int i = 0; i < 5; i++) { scene->objects[i] = walls->objects[i]; walls->objects[i] = nullptr; }
    delete walls;
    for (int32 i = 0; i < n_spheres; i++) {
        const uint64_t j = 7ULL * (uint64_t)i;
        uint32 col = 0xFF000000u;
        for (int c = 0; c < 3; c++) col |= (uint32)(64 + (int)std::floor(u(j + 4 + (uint64_t)c) * 192.0)) << (16 - 8 * c);
        scene->objects[5 + i] = new SphereObject(u(j) * 9.0 - 4.5, u(j + 1) * 9.0 - 4.5, u(j + 2) * 9.0 + 0.5, 0.02 + u(j + 3) * 0.06, col, 0.4, 0.0, 0.0, 1.0);
    }
    return scene;
}

Scene *loadScene(const char *file) {
    FILE *f = std::fopen(file, "rb");
    if (!f) { std::fprintf(stderr, "raytracer_amd: cannot open scene file %s\n", file); return nullptr; }
    std::vector<char> bytes;
    char buf[65536];
    for (size_t got; (got = std::fread(buf, 1, sizeof(buf), f)) > 0;) bytes.insert(bytes.end(), buf, buf + got);
    std::fclose(f);
    std::vector<pm_prim> prims;
    if (ends_with(file, ".bin")) {
        if (bytes.empty() || bytes.size() % sizeof(pm_prim)) { std::fprintf(stderr, "raytracer_amd: %s is not an array of %zu-byte primitive records\n", file, sizeof(pm_prim)); return nullptr; }
        prims.resize(bytes.size() / sizeof(pm_prim));
        std::memcpy(prims.data(), bytes.data(), bytes.size());
        for (size_t i = 0; i < prims.size(); i++)
            if (prims[i].kind != 0 && prims[i].kind != 1) { std::fprintf(stderr, "raytracer_amd: %s: record %zu is neither a sphere nor a plane\n", file, i); return nullptr; }
        return scene_from_records(prims);
    }
    bytes.push_back('\0');
    std::vector<SceneObject *> objs;
    int line_no = 0;
    for (char *line = bytes.data(), *next; line && *line; line = next) {
        next = std::strchr(line, '\n');
        if (next) *next++ = '\0';
        line_no++;
        if (char *hash = std::strchr(line, '#')) *hash = '\0';
        char what[16] = {0}, colour[32] = {0};
        double v[13] = {0};
        SceneObject *o = nullptr;
        if (std::sscanf(line, " %15s", what) != 1) continue;   // blank
        if (std::strcmp(what, "plane") == 0) {
            const int got = std::sscanf(line, " %*s %lf %lf %lf %lf %lf %31s %lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], &v[4], colour, &v[5], &v[6], &v[7], &v[8], &v[9]);
            if (got >= 10) { o = new PlaneObject(v[0], v[1], v[2], v[3], v[4], (uint32)std::strtoul(colour, nullptr, 16), v[5], v[6], v[7], v[8]); if (got >= 11) o->refraction = v[9]; }
        } else if (std::strcmp(what, "sphere") == 0) {
            const int got = std::sscanf(line, " %*s %lf %lf %lf %lf %31s %lf %lf %lf %lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], colour, &v[4], &v[5], &v[6], &v[7], &v[8], &v[9], &v[10], &v[11]);
            if (got >= 9) { o = new SphereObject(v[0], v[1], v[2], v[3], (uint32)std::strtoul(colour, nullptr, 16), v[4], v[5], v[6], v[7], got >= 13 ? v[9] : 0, got >= 13 ? v[10] : 0, got >= 13 ? v[11] : 0); if (got >= 10) o->refraction = v[8]; }
        }
        if (!o) {
            std::fprintf(stderr, "raytracer_amd: %s:%d: expected `plane x y z min max colour d s t a [refraction]` or `sphere x y z radius colour d s t a [refraction [dx dy dz]]`\n", file, line_no);
            for (SceneObject *x : objs) delete x;
            return nullptr;
        }
        objs.push_back(o);
    }
    if (objs.empty()) { std::fprintf(stderr, "raytracer_amd: %s describes no object\n", file); return nullptr; }
    Scene *scene = new Scene((int32)objs.size());
    for (size_t i = 0; i < objs.size(); i++) scene->objects[i] = objs[i];
    return scene;
}

void render(const char *image_file, int32 width, int32 height, int32 sample_ratio) { render(image_file, width, height, sample_ratio, nullptr); }

void render(const char *image_file, int32 width, int32 height, int32 sample_ratio, Scene *scene) {
    seed();
    std::printf("Setting up scene\n");
    if (!scene) scene = defaultScene();
    const int s = sample_ratio;
    Vec3 camera(0, 0, -12);
    // the supersampled pane (src/Image.cpp:32) stays on the GPU: the s x s box filter with integer division and the vertical
    // flip (src/Image.cpp:36-57) run there (pm_render_image), and only the width x height picture comes back
    std::vector<uint32> sample((size_t)width * height);
    if (width > 0 && height > 0 && s > 0) render_frame(scene, camera, nullptr, width * s, height * s, s, sample.data());
    delete scene;
    // the reference hands its little-endian 0xAARRGGBB words to stbi_write_png as RGBA bytes (src/Image.cpp:64),
    // so red and blue are swapped in its files; same bytes here
    if (!write_png(image_file, width, height, reinterpret_cast<const unsigned char *>(sample.data())))
        std::fprintf(stderr, "raytracer_amd: cannot write %s\n", image_file);
}

namespace backend {
bool writeImage(const char *file, int width, int height, const uint32 *words) { return write_png(file, width, height, reinterpret_cast<const unsigned char *>(words)); }
}

}  // namespace raytrace

namespace scheduler {
void startWorkers(int count) {
    std::printf("Starting GPU backend (MI355X); the %d CPU workers asked for are replaced by HIP kernels\n", count);
    (void)ctx();
}
// The reference queues one job per image row for its CPU workers (src/Raytrace.cpp:322-334).  Here renderScene IS the whole
// frame on the GPU, so a job a caller submits itself is host code of the caller's: it runs at once on the calling thread,
// in submission order, and waitForCompletion has the GPU work of the context to wait for, nothing else.
void submit(task job, void *job_data) { job(job_data); }
void waitForCompletion() { if (g_ctx) check(pm_ctx_synchronize(g_ctx), "pm_ctx_synchronize"); }
}  // namespace scheduler

namespace randutil {
void init(int64 s) { g_seed = (uint64_t)s; g_seeded = true; }
}

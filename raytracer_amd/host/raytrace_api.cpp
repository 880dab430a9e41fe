// raytrace_api.cpp -- host C++ API of the reference (see raytrace_api.h) above the C-ABI.
// No ray is traced and no photon is stored on the CPU here: every function forwards to
// libraytracer_amd.so and aborts with a message if the GPU backend reports an error.
#include "raytrace_api.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/raytracer_amd.h"

namespace {
pm_ctx *g_ctx = nullptr;
uint64_t g_seed = 0;
bool g_seeded = false;
int32 g_global_photons = 2048;    // NUM_PHOTONS, reference src/Raytrace.cpp:11
int32 g_caustic_photons = 2048;   // CAUSTIC_PHOTONS, reference src/Raytrace.cpp:19
const raytrace::Scene *g_uploaded = nullptr;

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
void upload(raytrace::Scene *scene) {
    std::vector<pm_prim> prims((size_t)scene->size);
    for (int i = 0; i < scene->size; i++) {
        std::memset(&prims[(size_t)i], 0, sizeof(pm_prim));
        if (!scene->objects[i]) { std::fprintf(stderr, "raytracer_amd: scene object %d is null\n", i); std::abort(); }
        scene->objects[i]->flatten(&prims[(size_t)i]);
    }
    check(pm_scene_upload(ctx(), prims.data(), scene->size), "pm_scene_upload");
    g_uploaded = scene;
}
uint64_t seed() {
    if (!g_seeded) {   // the reference seeds from the clock in raytrace::render (src/Image.cpp:16-17)
        g_seed = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        g_seeded = true;
    }
    return g_seed;
}
}  // namespace

namespace raytrace {

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
    if (g_uploaded == this) g_uploaded = nullptr;
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

static kdnode *make_map(int kind, int32 n, Vec3 &light_source, Vec3 &light_color, Scene *scene) {
    std::printf("Building %s photon map from %d photons\n", kind == PM_MAP_GLOBAL ? "global" : "caustic", n);
    auto t0 = std::chrono::high_resolution_clock::now();
    if (g_uploaded != scene) upload(scene);
    double lc[3] = {light_color.x, light_color.y, light_color.z};
    kdnode *node = new kdnode{nullptr, kind, n, nullptr};
    check(pm_map_create(ctx(), kind, n, seed(), lc, &node->map, nullptr), "pm_map_create");
    // the reference leaves the last emission point in light_source (src/PhotonMap.cpp:293); nothing reads it
    (void)light_source;
    std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
    std::printf("%s photons traced and indexed in %.3fs\n", kind == PM_MAP_GLOBAL ? "Global" : "Caustic", dt.count());
    return node;
}
kdnode *createPhotonMap(int32 n, Vec3 &light_source, Vec3 &light_color, Scene *scene) { return make_map(PM_MAP_GLOBAL, n, light_source, light_color, scene); }
kdnode *createCausticPhotonMap(int32 n, Vec3 &light_source, Vec3 &light_color, Scene *scene) { return make_map(PM_MAP_CAUSTIC, n, light_source, light_color, scene); }

int find_nearest_photons(photon **nearest, double *distances, int k, int size, Vec3 *target, kdnode *root, double max_dist) {
    if (k != PM_K || size != 0 || max_dist != 100) {
        std::fprintf(stderr, "raytracer_amd: find_nearest_photons supports k = 63, size = 0, max_dist = 100 (the reference's only call pattern)\n");
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

void renderScene(Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height) {
    Vec3 light_source(0, 4.96, 4);
    Vec3 light_color(0.6, 0.6, 0.6);
    kdnode *global_tree = createPhotonMap(g_global_photons, light_source, light_color, scene);
    kdnode *caustic_tree = createCausticPhotonMap(g_caustic_photons, light_source, light_color, scene);
    std::printf("Rendering scene\n");
    auto t0 = std::chrono::high_resolution_clock::now();
    pm_render_params rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.width = width; rp.height = height;
    rp.cam[0] = camera.x; rp.cam[1] = camera.y; rp.cam[2] = camera.z;
    rp.light_color[0] = light_color.x; rp.light_color[1] = light_color.y; rp.light_color[2] = light_color.z;
    rp.seed = seed(); rp.rank = 0; rp.world = 1;
    pm_stats st;
    check(pm_render(ctx(), global_tree->map, caustic_tree->map, &rp, pane, 0, &st), "pm_render");
    std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
    std::printf("Scene rendered in %.3fs\n", dt.count());
    deleteTree(global_tree);
    deleteTree(caustic_tree);
}

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

void render(const char *image_file, int32 width, int32 height, int32 sample_ratio) {
    seed();
    std::printf("Setting up scene\n");
    Scene *scene = new Scene(7);   // the Cornell box of src/Image.cpp:20-28
    scene->objects[0] = new PlaneObject(0, -5, 0, -5, 5, 0xFFEEEEEE, 0.4, 0.0, 0.0, 1.0);
    scene->objects[1] = new PlaneObject(0, 5, 0, -5, 5, 0xFDEEEEEE, 0.4, 0.0, 0.0, 1.0);   // the reference's 9-digit literal truncates to this
    scene->objects[2] = new PlaneObject(5, 0, 0, -5, 5, 0xFFFF3333, 0.4, 0.0, 0.0, 1.0);
    scene->objects[3] = new PlaneObject(-5, 0, 0, -5, 5, 0xFF3333FF, 0.4, 0.0, 0.0, 1.0);
    scene->objects[4] = new PlaneObject(0, 0, 10, -5, 5, 0xFFEEEEEE, 0.4, 0.0, 0.0, 1.0);
    scene->objects[5] = new SphereObject(2, -3.5, 3, 1.5, 0xFFFFFFFF, 0.0, 0.1, 0.9, 0.0);
    scene->objects[5]->refraction = 2.5;
    scene->objects[6] = new SphereObject(-2, -3.5, 5, 1.5, 0xFFFFFFFF, 0.0, 1.0, 0.0, 0.0);
    const int s = sample_ratio;
    std::vector<uint32> pane((size_t)width * height * s * s);
    Vec3 camera(0, 0, -12);
    renderScene(scene, camera, pane.data(), width * s, height * s);
    // s x s box filter with integer division, vertical flip (src/Image.cpp:36-57)
    std::vector<uint32> sample((size_t)width * height);
    for (int x = 0; x < width; x++)
        for (int y = 0; y < height; y++) {
            int32 red = 0, green = 0, blue = 0;
            for (int x0 = 0; x0 < s; x0++)
                for (int y0 = 0; y0 < s; y0++) {
                    uint32 v = pane[(size_t)(x * s + x0) + (size_t)(y * s + y0) * width * s];
                    red += (v >> 16) & 0xFF; green += (v >> 8) & 0xFF; blue += v & 0xFF;
                }
            red /= s * s; green /= s * s; blue /= s * s;
            sample[(size_t)x + (size_t)(height - y - 1) * width] = (0xFFu << 24) | ((uint32)red << 16) | ((uint32)green << 8) | (uint32)blue;
        }
    delete scene;
    // the reference hands its little-endian 0xAARRGGBB words to stbi_write_png as RGBA bytes (src/Image.cpp:64),
    // so red and blue are swapped in its files; same bytes here
    if (!write_png(image_file, width, height, reinterpret_cast<const unsigned char *>(sample.data())))
        std::fprintf(stderr, "raytracer_amd: cannot write %s\n", image_file);
}

}  // namespace raytrace

namespace scheduler {
void startWorkers(int count) {
    std::printf("Starting GPU backend (MI355X); the %d CPU workers asked for are replaced by HIP kernels\n", count);
    (void)ctx();
}
void submit(task job, void *job_data) { job(job_data); }
void waitForCompletion() {}
}  // namespace scheduler

namespace randutil {
void init(int64 s) { g_seed = (uint64_t)s; g_seeded = true; }
}

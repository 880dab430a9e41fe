// main.cpp -- the renderer's command line on the GPU backend.
//
//   raytracer [options] <cores> <width> <height> <samples>
//
// The four positional arguments and the three refusals are the reference's (src/Main.cpp:14-41): same wording, same
// exit status 0, nothing written.  Options the reference lacks (its photon count is a compile-time constant and its
// random engine is seeded from the clock):
//   --photons N   photons per map (also RAYTRACER_PHOTONS; default 2048, the reference's NUM_PHOTONS)
//   --seed S      fixes the random streams (also RAYTRACER_SEED; default: the clock, like the reference)
//   --out FILE    image file (default raytraced.png, the reference's name)
//   --scene WHAT  a scene other than the reference's Cornell box: a scene file (raytrace::loadScene: `*.bin` = pm_prim
//                 records, else text), `caustics` (BASELINE config 4), `spheres:N[:SEED]` (config 5: the walls + N spheres)
//   --rank R --world N --id-file PATH [--token T]
//                 one process per GPU (raytrace_dist.h): this process is rank R of N, each started with its own GPU visible
//                 (HIP_VISIBLE_DEVICES); rank 0 writes RCCL's unique id to PATH, the others wait for it; rank 0 writes the
//                 image.  Also RANK / WORLD_SIZE from the environment when --id-file is given.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytrace_api.h"
#include "raytrace_dist.h"

namespace {

struct Options {
    int cores = 1, width = 0, height = 0, samples = 0;
    long long photons = -1;
    bool have_seed = false;
    long long seed = 0;
    std::string out = "raytraced.png";
    std::string scene;
    int rank = 0, world = 0;          // world > 0: one process per GPU
    std::string id_file;
};

// One line per way the reference turns a command line down, in the order it checks them.
struct Refusal { bool (*applies)(const Options &); const char *message; };
const Refusal kRefusals[] = {
    {[](const Options &o) { return o.width / 16 != o.height / 9; }, "Image dimensions must be a 16:9 ratio.\n"},
    {[](const Options &o) { return o.samples < 0 || o.width < 0 || o.height < 0; }, "Dimensions and samples must be positive\n"},
};
const char kUsage[] = "Usage: ./raytracer [# cores] [width] [height] [samples]\n";

// Splits argv into options and the positional numbers; false = malformed option.
bool parse(int argc, char *argv[], Options &o, std::vector<const char *> &positional) {
    if (const char *e = std::getenv("RAYTRACER_SEED")) { o.have_seed = true; o.seed = std::atoll(e); }
    if (const char *e = std::getenv("RAYTRACER_PHOTONS")) o.photons = std::atoll(e);
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        const bool wants_value = a == "--photons" || a == "--seed" || a == "--out" || a == "--scene" || a == "--rank" || a == "--world" || a == "--id-file" || a == "--token";
        if (!wants_value) { positional.push_back(argv[i]); continue; }
        if (i + 1 >= argc) return false;
        const char *v = argv[++i];
        if (a == "--photons") o.photons = std::atoll(v);
        else if (a == "--seed") { o.have_seed = true; o.seed = std::atoll(v); }
        else if (a == "--scene") o.scene = v;
        else if (a == "--rank") o.rank = std::atoi(v);
        else if (a == "--world") o.world = std::atoi(v);
        else if (a == "--id-file") o.id_file = v;
        else if (a == "--token") setenv("RAYTRACER_GROUP_TOKEN", v, 1);     // what tells this launch's id file from a stale one (raytrace_dist.cpp)
        else o.out = v;
    }
    if (!o.id_file.empty() && o.world == 0) {
        if (const char *e = std::getenv("WORLD_SIZE")) o.world = std::atoi(e);
        if (const char *e = std::getenv("RANK")) o.rank = std::atoi(e);
    }
    if (o.world > 0 && (o.id_file.empty() || o.rank < 0 || o.rank >= o.world)) return false;
    return true;
}

}  // namespace

int main(int argc, char *argv[]) {
    Options o;
    std::vector<const char *> pos;
    if (!parse(argc, argv, o, pos) || pos.size() != 4) {
        std::fputs(kUsage, stdout);
        return 0;
    }
    o.cores = std::atoi(pos[0]) < 1 ? 1 : std::atoi(pos[0]);
    o.width = std::atoi(pos[1]); o.height = std::atoi(pos[2]); o.samples = std::atoi(pos[3]);
    for (const Refusal &r : kRefusals)
        if (r.applies(o)) { std::fputs(r.message, stdout); return 0; }
    // an option of ours that cannot work is NOT one of the reference's refusals (those print to stdout and return 0): it goes
    // to stderr with a non-zero exit code, so that a launcher can tell it from a frame that was rendered; checked before
    // anything touches the GPU
    long spheres_n = 0;
    unsigned long long spheres_seed = 2024ULL;
    if (o.scene.rfind("spheres:", 0) == 0) {
        char *end = nullptr;
        spheres_n = std::strtol(o.scene.c_str() + 8, &end, 10);
        if (end && *end == ':') spheres_seed = std::strtoull(end + 1, nullptr, 10);
        if (end == o.scene.c_str() + 8 || spheres_n < 0 || spheres_n > (1 << 22)) { std::fputs("--scene spheres:N needs 0 <= N <= 4194304\n", stderr); return 2; }
    }
    if (o.have_seed) randutil::init(o.seed);
    if (o.photons >= 0) raytrace::setPhotonCounts((int32)o.photons, (int32)o.photons);
    scheduler::startWorkers(o.cores);          // opens the GPU context; `cores` has no other meaning here
    raytrace::Scene *scene = nullptr;   // nullptr: the reference's Cornell box
    if (!o.scene.empty()) {
        if (o.scene == "caustics") scene = raytrace::causticsScene();
        else if (o.scene.rfind("spheres:", 0) == 0) {
            scene = raytrace::manySpheresScene((int32)spheres_n, spheres_seed);
        } else if (!(scene = raytrace::loadScene(o.scene.c_str()))) return 1;
    }
    if (o.world > 0) {
        raytrace::RankGroup *group = raytrace::joinRanks(o.rank, o.world, o.id_file.c_str());
        raytrace::renderRanks(group, o.out.c_str(), o.width, o.height, o.samples, scene);
        raytrace::leaveRanks(group);
    } else raytrace::render(o.out.c_str(), o.width, o.height, o.samples, scene);
    return 0;
}

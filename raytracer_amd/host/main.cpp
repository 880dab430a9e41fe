// main.cpp -- `raytracer cores width height samples`, the reference's command line (src/Main.cpp:14-41)
// on the GPU backend.  Environment (not in the reference): RAYTRACER_SEED fixes the random streams
// (default: clock, like the reference), RAYTRACER_PHOTONS sets the photons per map (default 2048).
#include <cstdio>
#include <cstdlib>

#include "raytrace_api.h"

int main(int argc, char *argv[]) {
    if (argc != 5) {
        printf("Usage: ./raytracer [# cores] [width] [height] [samples]\n");
        return 0;
    }
    int cores = atoi(argv[1]), width = atoi(argv[2]), height = atoi(argv[3]), samples = atoi(argv[4]);
    if (cores < 1) cores = 1;
    if (width / 16 != height / 9) {
        printf("Image dimensions must be a 16:9 ratio.\n");
        return 0;
    }
    if (samples < 0 || width < 0 || height < 0) {
        printf("Dimensions and samples must be positive\n");
        return 0;
    }
    if (const char *s = getenv("RAYTRACER_SEED")) randutil::init(atoll(s));
    if (const char *p = getenv("RAYTRACER_PHOTONS")) raytrace::setPhotonCounts(atoi(p), atoi(p));
    scheduler::startWorkers(cores);
    raytrace::render("raytraced.png", width, height, samples);
    return 0;
}

// raytrace_backend.h -- NOT part of the reference's API: the few pieces of raytrace_api.cpp's state that the multi-GPU frame
// (raytrace_dist.cpp) shares with it.
#pragma once
#include "raytrace_api.h"

struct pm_ctx;
namespace raytrace { namespace backend {
pm_ctx *context();                                   // the GPU context scheduler::startWorkers opened (opened now if it was not)
void uploadScene(Scene *scene);                      // flattened and sent when it differs from what the GPU holds
unsigned long long frameSeed();                      // randutil::init's seed, or the clock's as in src/Image.cpp:16-17
void photonCounts(int32 *global_photons, int32 *caustic_photons);
void checked(int rc, const char *what);              // aborts with the backend's message when rc != 0
bool writeImage(const char *file, int width, int height, const uint32 *words);   // PNG, the reference's byte order
} }

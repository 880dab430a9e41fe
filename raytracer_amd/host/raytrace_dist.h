// raytrace_dist.h -- renderScene for one process per GPU (reference src/Raytrace.cpp:301-343, split over ranks).
//
// The reference has one process and a pool of CPU threads.  Across the GPUs of a node the frame is split as BASELINE's
// north star says: emission attempts by contiguous ranges per rank, ONE RCCL all-gather of the ranks' photon shards per
// map, every rank builds the full index, image rows in interleaved blocks of 16 per rank, the blocks sent to rank 0.
// The maps -- and therefore the pane -- do not depend on the number of ranks.  This is the C++ twin of
// raytracer_amd/frame.py (which the bench and the tests drive through torch.distributed); it talks to RCCL directly.
//
//   rank 0..world-1, one process each, each with its own GPU visible as device 0 (HIP_VISIBLE_DEVICES) or
//   scheduler::startWorkers called after hipSetDevice; id_file: a path all ranks can read -- rank 0 writes RCCL's
//   unique id there, the others wait for it.  The file must not exist when the ranks start (a stale one would be read as
//   the id); rank 0 removes it in leaveRanks.
#pragma once
#include "raytrace_api.h"

namespace raytrace {

struct RankGroup;
RankGroup *joinRanks(int32 rank, int32 world, const char *id_file);
void leaveRanks(RankGroup *group);
// pane (width x height words) is filled on rank 0 only
void renderSceneRanks(RankGroup *group, Scene *scene, Vec3 &camera, uint32 *pane, int32 width, int32 height);
// raytrace::render on the group: every rank traces its rows of the supersampled pane, rank 0 filters and writes the file
void renderRanks(RankGroup *group, const char *image_file, int32 width, int32 height, int32 sample_ratio, Scene *scene);

}  // namespace raytrace

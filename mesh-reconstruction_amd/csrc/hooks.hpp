// hooks.hpp -- every environment switch of libmvs_hip.so, in one place, behind ONE master switch.
//
// A production process sets none of them: unless MVS_TEST_HOOKS=1 is in the environment the library reads no other variable and every
// field below keeps its default.  The environment is looked at when a context or a communicator is CREATED (mvs_create,
// mvs_comm_create; the context-free Poisson entry once per process) and never on a per-call path: what a context was created under is
// what it keeps (the snapshot lives in mvs_ctx / mvs_comm).  The table in INTEGRATION.md section 7 lists each variable with what it is for.
#pragma once

#include <string>

namespace mvs {

struct Hooks {
    bool enabled = false;              // MVS_TEST_HOOKS=1: the variables below are honoured at all
    // ---- multi-GPU (csrc/comm.cpp) ----
    std::string rccl_library;          // MVS_RCCL_LIBRARY: load this file instead of librccl.so.1 (tests/loopback_rccl: n ranks on one GPU)
    bool comm_allow_same_device = false;  // MVS_COMM_ALLOW_SAME_DEVICE=1: a device may be listed more than once
    int comm_fail_rank = -1;           // MVS_COMM_TEST_FAIL_RANK=r: rank r gives up in its local phase, as a failed allocation would
    bool comm_allreduce = false;       // MVS_COMM_ALLREDUCE=1: MVS_SHARD_VIEWS_SCATTER runs as the all-reduce pipeline
    // ---- sweep planners and kernels ----
    bool debug_flags = false;          // MVS_DEBUG_FLAGS=1: bits 8-23 of mvs_sweep_run's flags (timing experiments, forced splits) are honoured
    bool no_rect = false;              // MVS_NO_RECT=1: never plan the rectified-view kernels
    bool no_plan_cache = false;        // MVS_NO_PLAN_CACHE=1: initial state of mvs_sweep_set_plan_cache (0 = plan every view set)
    std::string plan_dump;             // MVS_PLAN_DUMP=<file>: plan_regions_fx writes its descriptors there (tools/plan_hist.py)
    bool rect_verbose = false;         // MVS_RECT_VERBOSE=1: box sizes of the rectified planners on stderr
    // ---- other stages ----
    bool filter_timing = false;        // MVS_FILTER_TIMING=1: stage timer of mvs_filter_points on stderr
    int filter_sorted_lists = -1;      // MVS_FILTER_SORTED_LISTS=0|1: never / always the global-sort path of the neighbour lists
    int filter_max_rounds = 2048;      // MVS_FILTER_MAX_ROUNDS=<n>: greedy rounds on the device before the host finishes the walk
    bool serial_flows = false;         // MVS_SERIAL_FLOWS=1: mvs_process_frame runs its flows on the main stream (A/B of the lanes)
    bool fb_lanes = false;             // MVS_FB_LANES=1: Farneback flows one chain per side view on the lanes (A/B of the batched pass)
    bool fb_unfused = false;           // MVS_FB_UNFUSED=1: Farneback iteration as three kernels
    bool fb_serial_prep = false;       // MVS_FB_SERIAL_PREP=1: Farneback's pyramid preparation level by level (3 launches per level) instead of all levels in 3 launches (A/B of round 6's form)
    bool var_unfused = false;          // MVS_VAR_UNFUSED=1: variational fixed-point iteration as separate kernels
    bool fb_direct_box = false;        // MVS_FB_DIRECT_BOX=1: the fused Farneback iteration sums its window term by term (round 2-4's kernel)
    bool flow_graph = false;           // MVS_FLOW_GRAPH=1: mvs_flow replays its kernel sequence as a hipGraph, as rounds 2-3 did (tools/graph_repro.py: the
                                       // replay-after-first-Poisson-call corruption of round 4; never set otherwise)
    bool flow_graph_kernel_memset = false;  // MVS_FLOW_GRAPH=2: the same, with the sequence's hipMemsetAsync calls replaced by a zero-fill kernel (the A/B of the finding)
    bool no_sep = false;               // MVS_NO_SEP=1: the general tiled kernel never takes its separable path (A/B: bit-identical)
    int onecall_bands = 0;             // MVS_ONECALL_BANDS=n: row bands of the one-call mvs_sweep's upload pipeline (default: 2 for view sets of 16 MB and more; 1 = the unbanded path)
    int onecall_first_permille = 0;    // MVS_ONECALL_FIRST=p: with two bands, the first one takes p/1000 of the rows (timing A/B)
    int fb_variant = 0;                // MVS_FB_VARIANT=2: the tall Farneback tiles with 512 threads x 4 rows instead of 256 x 8 (timing A/B: slower)
    int raster_bins = -1;              // MVS_RASTER_BINS=0|1: never / always bin the faces per tile
    bool poison_alloc = false;         // MVS_POISON_ALLOC=1: fresh device allocations are filled with 0xFF bytes
};

// reads the environment (getenv): call when a context / communicator is created, keep the result
Hooks read_hooks();
// the snapshot the process took at its first call of this function (context-free entry points: the Poisson mesher's allocator)
const Hooks &process_hooks();

}  // namespace mvs

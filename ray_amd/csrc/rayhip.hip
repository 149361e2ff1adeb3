// rayhip.hip -- implementation of the librayhip C ABI (include/rayhip.h): device memory, stage schedule, readback.
//
// Stage schedule of one rayhip_render == one RenderScene iteration, restating the control flow of
// reference internal/RendererCPU.h:373-659 with the GPU-side bookkeeping of internal/RendererVK.cpp:368-791
// (everything on one stream, ray counts never leave HBM).
//
// There is NO host fallback in this library: without a gfx950 device every entry point fails.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rayhip.h"
#include "kernels.hip.h" // first: it configures the profiling macros the rt_*.h headers expand
#include "shade_launch.h"
#include "bvh4_build.h"
#include "bvh4_build.hip.h"
#include "bvh8_build.h"
#include "bvh_layout.h"
#include "unet.h"
#include "lbvh.hip.h"
#include "scene_blob.h"
#include "scene_rebuild.h"
#include "scene_update.h"
#include "scene_validate.h"
#include "sort.h"

using namespace rt;

// the host side, in parts (one translation unit with the kernels above: templates and launch sites see each other)
#include "rayhip_ctx.hip.h"
#include "rayhip_upload.hip.h"
#include "rayhip_render.hip.h"
#include "rayhip_denoise.hip.h"
#include "rayhip_frames.hip.h"
#include "rayhip_hooks.hip.h"
#include "comm.hip.h"

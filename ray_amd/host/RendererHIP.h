// RendererHIP.h -- the new backend's factory, in the style of the reference's internal/RendererVK.h:9-10.
//
// Drop-in recipe (INTEGRATION.md has the full patch):
//   RendererBase.h:22-34   enum class eRendererType { ..., Vulkan, DirectX12, HIP };   // + RendererGPU mask :41
//   RendererBase.cpp:6-48  "HIP" <-> eRendererType::HIP in RendererTypeName / RendererTypeFromName
//   Ray.cpp:53-73          if (enabled_types & eRendererType::HIP) try { return Hip::CreateRenderer(s, log); } catch ...
//   Config.h.in            #cmakedefine ENABLE_HIP_IMPL
// This tree cannot edit the (read-only) reference, so the type id is provided here as a constant with the
// value the enum entry would get.
#pragma once

#include <cstdint>
#include <vector>

#include "RendererBase.h"

namespace Ray {
class ILog;
namespace Hip {
// value of eRendererType::HIP once appended after DirectX12 (RendererBase.h:33)
constexpr eRendererType RendererTypeHIP = eRendererType(uint32_t(eRendererType::DirectX12) + 1);

// Throws std::runtime_error when no gfx950 device / librayhip is unavailable -- the factory convention of the
// GPU backends (Ray.cpp:58-63) -- so that Ray::CreateRenderer can fall through to the next enabled type.
RendererBase *CreateRenderer(const settings_t &s, ILog *log);

// SceneHIP without a renderer (scene construction is host-only work) and its flat serialisation
// (ray_amd/csrc/scene_blob.h): what rayhip_scene_upload_blob consumes.
SceneBase *CreateScene(ILog *log, bool use_tex_compression = false);
std::vector<uint8_t> ExportSceneBlob(const SceneBase &scene);
// where the last Finalize baked the sky environment map of a SceneHIP: "device" (rayhip_bake_sky, a scene made by a renderer), "host" (the
// reference's loop: a scene without a renderer) or "none" (no physical sky)
const char *SkyBakedOn(const SceneBase &scene);
} // namespace Hip
} // namespace Ray

// The sanitizer build holds no kernel translation unit: a decoder set created with PHQ_DEVICE_NONE never launches, and
// these two entry points of the stage dispatcher (stage.hip) only have to link.
#include "../device_types.hpp"

namespace phq {
int launch_stage(const StageArgs&, const DevDecoder&, int32_t, int32_t, void*, LaunchTimer*) { return 100; }      // hipErrorNoDevice
bool stage_uses_worklist(const DevDecoder&, int32_t) { return false; }
}

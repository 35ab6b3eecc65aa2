// The sanitizer build holds no kernel translation unit: a decoder set created with PHQ_DEVICE_NONE never launches, and
// these entry points of the kernel translation units (stage.hip, fastq_device.hip) only have to link.
#include "../device_types.hpp"
#include "../fastq_device.hpp"
#include "../record_device.hpp"
#include "../unpack.hpp"

namespace phq {
int launch_stage(const StageArgs&, const DevDecoder&, int32_t, int32_t, void*, LaunchTimer*) { return 100; }      // hipErrorNoDevice
bool stage_uses_worklist(const DevDecoder&, int32_t) { return false; }
void stage_plan_only(bool) {}
// ... and the device FASTQ path (fastq_device.hip) is never taken by a set without a device
int unpack_planes(const uint8_t*, uint8_t*, uint8_t*, size_t, void*) { return 100; }
int patch_escapes(const uint64_t*, int64_t, int64_t, int64_t, int32_t, const int32_t*, const int32_t*, uint8_t*, uint8_t*, void*) { return 100; }
int record_sizes_by_sink(const FastqTables*, const FastqPlan&, int32_t, uint32_t, int64_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint32_t*, const RecordRound&, int32_t, int32_t, void*) { return 100; }
int record_write_by_sink(const FastqTables*, const FastqPlan&, int32_t, uint32_t, int64_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint32_t*, const RecordRound&, int32_t, int32_t, void*) { return 100; }
int record_quality_control(const FastqTables*, const FastqPlan&, int32_t, uint32_t, int64_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint32_t*, const RecordRound&, uint32_t*, void*) { return 100; }
void fastq_plan_waves(FastqPlan&, double) {}
void fastq_fill_tables(FastqTables&) {}
int fastq_index(const FastqFeedChunk&, void*) { return 100; }
int fastq_plan_round(const FastqFeedChunk*, int32_t, const int32_t*, uint32_t, FastqRound*, void*) { return 100; }
int fastq_decode(const FastqTables*, const FastqPlan&, int32_t, uint32_t, int64_t, uint8_t*, uint8_t*, uint8_t*, uint32_t*, unsigned long long*, uint8_t*, uint32_t*, uint32_t*, uint32_t*, void*) { return 100; }
int record_sizes(const FastqTables*, const FastqPlan&, int32_t, uint32_t, int64_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint32_t*, const RecordRound&, void*) { return 100; }
int record_write(const FastqTables*, const FastqPlan&, int32_t, uint32_t, int64_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint32_t*, const RecordRound&, void*) { return 100; }
}

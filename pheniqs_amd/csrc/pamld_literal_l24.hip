// PAMLD literal kernels for nucleotide cardinality 17..24 (split by range so the ranges build in parallel).
#include "pamld_literal.hpp"

namespace phq {
int launch_pamld_literal_l24(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    return launch_pamld_literal_range<17, 24>(args, host_decoder, grid, lds_bytes, stream);
}
}  // namespace phq

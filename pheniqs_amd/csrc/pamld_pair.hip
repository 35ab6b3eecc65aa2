// Paired barcode tier of the PAMLD path (pamld_pair.hpp): two segments of up to 10 nt, one barcode scored.
#include <cstdlib>

#include "pamld_pair.hpp"

namespace phq {

bool pamld_pair_supports(const DevDecoder& host_decoder) {
    const int words = host_decoder.plane_stride / 4;
    return host_decoder.algorithm == 1 && host_decoder.pair_ready != 0 && (host_decoder.use_lds_histogram != 0 || count_results_supports(host_decoder)) &&
           words >= 2 && words <= 8 && host_decoder.nucleotide_cardinality <= 32 &&
           (pamld_lut_supports(host_decoder) || pamld_wide_supports(host_decoder));
}

// the second look inside the launch (LISTS): the lists are there and the waves' rings fit the workgroup's LDS beside the records
// and the histogram, several workgroups a CU.  Round 6: sets of more than 512 barcodes take up to 80 KiB for it — two workgroups
// a CU; their product form behind the kernel is dear enough for the second look to pay whatever the occupancy (1 536 pairs of
// 10 + 10 nt 4.08 -> 2.88 ms per 16 M reads, 768 pairs of 12 + 12 nt 2.07 -> 1.75) — while sets of 192 and 288 pairs with their
// histogram in LDS lose 5 to 19 % the same way and keep the 40 KiB rule (profiles/tools/scratch/pair_lists_room.sh)
template <int G>
bool lists_fit(const DevDecoder& host_decoder, size_t histogram_bytes) {
    static const size_t forced = []() { const char* text = std::getenv("PHQ_PAIR_LISTS_LDS"); return text ? static_cast<size_t>(std::atol(text)) : size_t(0); }();      // (the variable: sweeps)
    const size_t bytes = pair::LayoutOf<G, true>::histogram_at(host_decoder.n_barcode) + histogram_bytes;
    if(forced) { return bytes <= forced; }
    return bytes <= 40u * 1024u || (host_decoder.n_barcode > 512 && bytes <= 80u * 1024u);
}

int launch_pamld_pair(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    static const bool no_lists = std::getenv("PHQ_NO_PAIR_LISTS") != nullptr;
    const bool lists = !no_lists && host_decoder.pair_list != nullptr;
    switch(host_decoder.plane_stride / 4) {
        case 2: return pair::launch_one<2>(args, host_decoder.n_barcode, lists && lists_fit<2>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        case 3: return pair::launch_one<3>(args, host_decoder.n_barcode, lists && lists_fit<3>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        case 4: return pair::launch_one<4>(args, host_decoder.n_barcode, lists && lists_fit<4>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        case 5: return pair::launch_one<5>(args, host_decoder.n_barcode, lists && lists_fit<5>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        case 6: return pair::launch_one<6>(args, host_decoder.n_barcode, lists && lists_fit<6>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        case 7: return pair::launch_one<7>(args, host_decoder.n_barcode, lists && lists_fit<7>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        case 8: return pair::launch_one<8>(args, host_decoder.n_barcode, lists && lists_fit<8>(host_decoder, histogram_bytes), grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq

// Host-only self check of the library's CPU side, built with AddressSanitizer + UBSan (make -C pheniqs_amd/csrc asan):
// JSON reader, configuration compiler, compiled job model, PhredScale, statistics model, report writer.  GPU sanitizers are not available
// on the target pool, so the device code is checked by the oracle parity tests instead.
//
// usage: host_check <compiled job json> [more jobs ...]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

#include "../compile.hpp"
#include "../error.hpp"
#include "../job.hpp"
#include "../json.hpp"
#include "../phred.hpp"
#include "../stats.hpp"
#include "../text_format.hpp"
#include "../url.hpp"

static int failures = 0;
#define CHECK(condition) do { if(!(condition)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #condition); ++failures; } } while(0)

int main(int argc, char** argv) {
    using namespace phq;
    const PhredScale& scale = PhredScale::get_instance();
    CHECK(scale.substitution_quality(1, 1, 30) == scale.true_positive_quality[30]);
    CHECK(scale.substitution_quality(1, 2, 30) == 30.0);
    CHECK(scale.substitution_quality(1, 15, 30) == scale.uniform_base_quality);
    CHECK(scale.substitution_quality(1, 1, 0) == 0.0);
    CHECK(std::fabs(scale.uniform_probability - 0.25) < 1e-15);
    CHECK(format_double(0.18 * (1 - 0.015), 15) == "0.177299999999999");
    CHECK(format_double(1.0 / 65536, 15) == "0.000015258789062");
    CHECK(format_double(248.0, 15) == "248.0");

    // the SAM float tag writer against printf("%g"): random magnitudes, neighbours of short decimals, sixth digit halves
    {
        uint64_t state = 88172645463325252ull;
        auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
        auto same = [](float value) {
            char mine[64], theirs[64];
            *put_float_g(mine, value) = 0;
            std::snprintf(theirs, sizeof(theirs), "%g", static_cast<double>(value));
            return std::string(mine) == theirs;
        };
        for(int i = 0; i < 400000; ++i) {
            const double exponent = -24.0 * static_cast<double>(next() >> 11) / 9007199254740992.0;
            CHECK(same(static_cast<float>(std::pow(10.0, exponent))));
        }
        for(int k = -21; k <= 0; ++k) {
            for(int m = 1; m < 300; ++m) {
                const float value = static_cast<float>(m * std::pow(10.0, k - 2));
                CHECK(same(value)); CHECK(same(std::nextafterf(value, 0.0f))); CHECK(same(std::nextafterf(value, 2.0f)));
            }
            const float half = static_cast<float>(100000.5 * std::pow(10.0, k - 5));
            CHECK(same(half)); CHECK(same(std::nextafterf(half, 0.0f))); CHECK(same(std::nextafterf(half, 2.0f)));
        }
        CHECK(same(0.999999f)); CHECK(same(0.99999994f)); CHECK(same(1e-4f)); CHECK(same(9.99999e-5f)); CHECK(same(1.0f)); CHECK(same(0.0f)); CHECK(same(1e-30f));
    }
    // file references: extension and query (url.cpp:280-417)
    {
        const Url a = Url::parse("out/reads_s1.fastq.gz");
        CHECK(a.format == FileFormat::FASTQ && a.compression() == FileCompression::GZIP && a.path == "out/reads_s1.fastq.gz");
        const Url b = Url::parse("/dev/stdout?format=sam&compression=none");
        CHECK(b.format == FileFormat::SAM && b.compression() == FileCompression::NONE && b.is_stdout() && b.path == "/dev/stdout");
        Url c = Url::parse("x.fq?compression=gz&level=7&unknown=1");
        CHECK(c.format == FileFormat::FASTQ && c.compression() == FileCompression::GZIP && c.compression_level == 7);
        c.relocate_child("base/dir");
        CHECK(c.path == "base/dir/x.fq" && c.encoded() == "base/dir/x.fq?compression=gz&level=7&unknown=1");
        Url d = Url::parse("/abs/y.bam");
        d.relocate_child("base");
        CHECK(d.path == "/abs/y.bam" && d.format == FileFormat::BAM);
        CHECK(Url::parse("plain").format == FileFormat::UNKNOWN && Url::parse("a.b").format == FileFormat::UNKNOWN);
    }

    // malformed documents must throw ConfigurationError, never crash
    const char* broken[] = {"", "{", "{\"a\":", "[1,2", "{\"a\" 1}", "\"unterminated", "{\"sample\": {\"algorithm\": 3}}", "nul", "{\"a\": 1} trailing"};
    for(const char* text : broken) {
        bool threw = false;
        try { JobConfig::from_json(Json::parse(text)); } catch(const Error&) { threw = true; }
        CHECK(threw || std::string(text) == "{\"a\": 1} trailing");
    }

    for(int i = 1; i < argc; ++i) {
        std::ifstream file(argv[i]);
        std::stringstream content;
        content << file.rdbuf();
        try {
            Json document = Json::parse(content.str());
            if(!is_compiled_job(document)) {
                document = compile_job(document);                      // a user level document: repository, bases, defaults
                CHECK(is_compiled_job(document));
                CHECK(Json::parse(document.dump(324)).dump(324) == document.dump(324));
            }
            const JobConfig job = JobConfig::from_json(document);
            CHECK(!job.decoder.empty());
            // a deterministic pseudo random counter buffer through decode / finalize / encode
            std::vector<uint64_t> words(static_cast<size_t>(job.stats_words));
            uint64_t state = 88172645463325252ull;
            for(auto& w : words) { state ^= state << 13; state ^= state >> 7; state ^= state << 17; w = state % 100000; }
            const Json report = encode_report(job, words.data(), 1000000, 990000);
            const std::string text = report.dump(job.float_precision);
            const Json again = Json::parse(text);
            CHECK(again.is_object() && !again.members().empty());
            // and the all-zero buffer (every guard against division by zero, selector.cpp:78-101,197-214)
            std::vector<uint64_t> zero(static_cast<size_t>(job.stats_words), 0);
            CHECK(!encode_report(job, zero.data(), 0, 0).dump(15).empty());
            std::printf("%s: %zu decoders, %lld counter words, report %zu bytes\n", argv[i], job.decoder.size(),
                        static_cast<long long>(job.stats_words), text.size());
            // the same document with bytes flipped, dropped or doubled: a configuration error or a job, never a crash
            // (sanitizers watching); structural damage is the common case, valid JSON with a wrong type the other one
            {
                const std::string original = content.str();
                uint64_t fuzz = 1442695040888963407ull + static_cast<uint64_t>(i);
                auto next = [&]() { fuzz ^= fuzz << 13; fuzz ^= fuzz >> 7; fuzz ^= fuzz << 17; return fuzz; };
                int accepted = 0;
                for(int round = 0; round < 400; ++round) {
                    std::string mutated = original;
                    const int edits = 1 + static_cast<int>(next() % 3);
                    for(int e = 0; e < edits && !mutated.empty(); ++e) {
                        const size_t at = static_cast<size_t>(next() % mutated.size());
                        switch(next() % 5) {
                            case 0: mutated[at] = static_cast<char>(next() % 96 + 32); break;
                            case 1: mutated.erase(at, 1 + static_cast<size_t>(next() % 8)); break;
                            case 2: mutated.insert(at, mutated.substr(at, 1 + static_cast<size_t>(next() % 16))); break;
                            case 3: mutated[at] = "{}[],:\"0-9nulltruefalse"[next() % 24]; break;
                            default: mutated.replace(at, std::min<size_t>(4, mutated.size() - at), "null"); break;
                        }
                    }
                    try {
                        Json damaged = Json::parse(mutated);
                        if(!is_compiled_job(damaged)) { damaged = compile_job(damaged); }
                        const JobConfig survivor = JobConfig::from_json(damaged);
                        std::vector<uint64_t> none(static_cast<size_t>(survivor.stats_words), 0);
                        (void)encode_report(survivor, none.data(), 0, 0).dump(15);
                        ++accepted;
                    } catch(const Error&) {
                    } catch(const std::exception&) {
                    }
                }
                std::printf("%s: 400 damaged copies, %d still a job\n", argv[i], accepted);
            }
        } catch(const Error& e) {
            std::fprintf(stderr, "%s: %s\n", argv[i], e.what());
            ++failures;
        }
    }
    if(failures) { std::fprintf(stderr, "%d failure(s)\n", failures); return 1; }
    std::puts("host check ok");
    return 0;
}

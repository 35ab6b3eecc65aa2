// File references of a job document: `path?format=fastq&compression=gz`.
//
//   URL::parse                 url.cpp:280-357    query after '?', implicit type and compression from the extensions
//   URL::apply_query_parameter url.cpp:393-417    format / compression / level; unknown keys ignored
//   URL::relocate_child        url.cpp:531-544    a relative path goes under the base directory
//
// Only what the FASTQ / SAM text feeds of the transcoder need; BAM and CRAM are recognised so that they can be refused
// by name (they need htslib).
#pragma once

#include <string>

namespace phq {

enum class FileFormat : int { UNKNOWN = 0, NONE, FASTQ, SAM, BAM, CRAM, JSON };
enum class FileCompression : int { UNKNOWN = 0, NONE, GZIP, BGZF, BZ2, XZ };

struct Url {
    std::string path;
    std::string query;
    FileFormat format = FileFormat::UNKNOWN;
    FileCompression implicit_compression = FileCompression::UNKNOWN;       // from a trailing .gz / .bz2 / .xz
    FileCompression explicit_compression = FileCompression::UNKNOWN;       // from the query
    int compression_level = -1;

    static FileFormat format_of(const std::string& name) {                 // url.cpp:86-97
        if(name == "none") { return FileFormat::NONE; }
        if(name == "fastq" || name == "fq") { return FileFormat::FASTQ; }
        if(name == "sam") { return FileFormat::SAM; }
        if(name == "bam") { return FileFormat::BAM; }
        if(name == "cram") { return FileFormat::CRAM; }
        if(name == "json") { return FileFormat::JSON; }
        return FileFormat::UNKNOWN;
    }
    static FileCompression compression_of(const std::string& name) {       // url.cpp:144-152
        if(name == "none") { return FileCompression::NONE; }
        if(name == "gz") { return FileCompression::GZIP; }
        if(name == "bgzf") { return FileCompression::BGZF; }
        if(name == "bz2") { return FileCompression::BZ2; }
        if(name == "xz") { return FileCompression::XZ; }
        return FileCompression::UNKNOWN;
    }

    static Url parse(const std::string& encoded) {
        Url url;
        const size_t mark = encoded.find('?');
        url.path = encoded.substr(0, mark);
        if(mark != std::string::npos) { url.query = encoded.substr(mark + 1); }
        const size_t slash = url.path.find_last_of('/');
        std::string name = slash == std::string::npos ? url.path : url.path.substr(slash + 1);
        size_t dot = name.find_last_of('.');
        if(dot != std::string::npos && dot > 0 && dot + 2 < name.size()) {
            std::string extension = name.substr(dot + 1);
            name.erase(dot);
            if(extension == "gz" || extension == "bz2" || extension == "xz") {
                url.implicit_compression = compression_of(extension);
                extension.clear();
                dot = name.find_last_of('.');
                if(dot != std::string::npos && dot + 2 < name.size()) { extension = name.substr(dot + 1); }
            }
            if(!extension.empty()) { url.format = format_of(extension); }
        }
        size_t at = 0;
        while(at < url.query.size()) {
            size_t end = url.query.find('&', at);
            if(end == std::string::npos) { end = url.query.size(); }
            const std::string pair = url.query.substr(at, end - at);
            const size_t equals = pair.find('=');
            if(equals != std::string::npos && equals > 0 && equals + 1 < pair.size()) {
                const std::string key = pair.substr(0, equals);
                const std::string value = pair.substr(equals + 1);
                if(key == "format") { url.format = format_of(value); }
                else if(key == "compression") { url.explicit_compression = compression_of(value); }
                else if(key == "level") { url.compression_level = value[0] >= '0' && value[0] <= '9' ? value[0] - '0' : -1; }
            }
            at = end + 1;
        }
        return url;
    }

    bool is_absolute() const { return !path.empty() && path[0] == '/'; }
    bool is_stdin() const { return path == "/dev/stdin" || path == "-"; }
    bool is_stdout() const { return path == "/dev/stdout"; }
    bool is_stderr() const { return path == "/dev/stderr"; }
    bool is_null() const { return path == "/dev/null"; }
    bool is_standard_stream() const { return is_stdin() || is_stdout() || is_stderr() || is_null(); }
    FileCompression compression() const {
        return explicit_compression != FileCompression::UNKNOWN ? explicit_compression : implicit_compression;
    }
    void relocate_child(const std::string& base) {
        if(base.empty() || is_absolute() || path == "-") { return; }
        path = base.back() == '/' ? base + path : base + "/" + path;
    }
    std::string encoded() const { return query.empty() ? path : path + "?" + query; }
};

}  // namespace phq

#include "json.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "error.hpp"

namespace phq {

namespace {

struct Parser {
    const char* p;
    const char* end;
    const char* begin;

    [[noreturn]] void fail(const std::string& what) const {
        throw ConfigurationError(what + " at position " + std::to_string(p - begin));
    }
    void skip() {
        while(p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) { ++p; }
    }
    Json value() {
        skip();
        if(p >= end) { fail("unexpected end of JSON document"); }
        switch(*p) {
            case '{': return object();
            case '[': return array();
            case '"': return Json::string(string());
            case 't': literal("true"); return Json::boolean(true);
            case 'f': literal("false"); return Json::boolean(false);
            case 'n': literal("null"); return Json();
            default: return number();
        }
    }
    void literal(const char* word) {
        size_t n = std::strlen(word);
        if(static_cast<size_t>(end - p) < n || std::strncmp(p, word, n) != 0) { fail("invalid JSON literal"); }
        p += n;
    }
    Json number() {
        const char* start = p;
        bool real = false;
        if(p < end && (*p == '-' || *p == '+')) { ++p; }
        while(p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
            if(*p == '.' || *p == 'e' || *p == 'E') { real = true; }
            ++p;
        }
        if(p == start) { fail("invalid JSON value"); }
        std::string token(start, p);
        if(real) {
            char* tail = nullptr;
            double v = std::strtod(token.c_str(), &tail);
            if(tail == token.c_str()) { fail("invalid JSON number"); }
            return Json::real(v);
        }
        if(token[0] == '-') {
            return Json::integer(std::strtoll(token.c_str(), nullptr, 10));
        }
        unsigned long long u = std::strtoull(token.c_str(), nullptr, 10);
        if(u > static_cast<unsigned long long>(INT64_MAX)) { return Json::uinteger(u); }
        return Json::integer(static_cast<int64_t>(u));
    }
    static void utf8(std::string& out, unsigned cp) {
        if(cp < 0x80) { out.push_back(static_cast<char>(cp)); }
        else if(cp < 0x800) { out.push_back(static_cast<char>(0xC0 | (cp >> 6))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
        else if(cp < 0x10000) {
            out.push_back(static_cast<char>(0xE0 | (cp >> 12)));
            out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else {
            out.push_back(static_cast<char>(0xF0 | (cp >> 18)));
            out.push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        }
    }
    unsigned hex4() {
        if(end - p < 4) { fail("truncated unicode escape"); }
        unsigned v = 0;
        for(int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if(c >= '0' && c <= '9') { v |= static_cast<unsigned>(c - '0'); }
            else if(c >= 'a' && c <= 'f') { v |= static_cast<unsigned>(c - 'a' + 10); }
            else if(c >= 'A' && c <= 'F') { v |= static_cast<unsigned>(c - 'A' + 10); }
            else { fail("invalid unicode escape"); }
        }
        return v;
    }
    std::string string() {
        ++p;  // opening quote
        std::string out;
        while(true) {
            if(p >= end) { fail("unterminated string"); }
            char c = *p++;
            if(c == '"') { break; }
            if(c == '\\') {
                if(p >= end) { fail("unterminated escape"); }
                char e = *p++;
                switch(e) {
                    case '"': out.push_back('"'); break;
                    case '\\': out.push_back('\\'); break;
                    case '/': out.push_back('/'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break;
                    case 't': out.push_back('\t'); break;
                    case 'u': {
                        unsigned cp = hex4();
                        if(cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            p += 2;
                            unsigned low = hex4();
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (low - 0xDC00);
                        }
                        utf8(out, cp);
                        break;
                    }
                    default: fail("invalid escape");
                }
            } else { out.push_back(c); }
        }
        return out;
    }
    Json array() {
        ++p;
        Json out = Json::array();
        skip();
        if(p < end && *p == ']') { ++p; return out; }
        while(true) {
            out.push(value());
            skip();
            if(p >= end) { fail("unterminated array"); }
            if(*p == ',') { ++p; continue; }
            if(*p == ']') { ++p; break; }
            fail("expected , or ] in array");
        }
        return out;
    }
    Json object() {
        ++p;
        Json out = Json::object();
        skip();
        if(p < end && *p == '}') { ++p; return out; }
        while(true) {
            skip();
            if(p >= end || *p != '"') { fail("expected a member name"); }
            std::string key = string();
            skip();
            if(p >= end || *p != ':') { fail("expected : after member name"); }
            ++p;
            out.set(key, value());
            skip();
            if(p >= end) { fail("unterminated object"); }
            if(*p == ',') { ++p; continue; }
            if(*p == '}') { ++p; break; }
            fail("expected , or } in object");
        }
        return out;
    }
};

void escape(std::string& out, const std::string& s) {
    out.push_back('"');
    for(unsigned char c : s) {
        switch(c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if(c < 0x20) {
                    static const char* hex = "0123456789ABCDEF";
                    out += "\\u00";
                    out.push_back(hex[c >> 4]);
                    out.push_back(hex[c & 0xF]);
                } else { out.push_back(static_cast<char>(c)); }
        }
    }
    out.push_back('"');
}

}  // namespace

Json Json::parse(const std::string& text) {
    Parser parser{text.data(), text.data() + text.size(), text.data()};
    Json out = parser.value();
    parser.skip();
    if(parser.p != parser.end) { parser.fail("trailing characters after JSON document"); }
    return out;
}

const Json* Json::find(const std::string& key) const {
    if(type_ != Type::Object) { return nullptr; }
    for(const auto& member : members_) {
        if(member.first == key) { return &member.second; }
    }
    return nullptr;
}

void Json::set(const std::string& key, Json value) {
    for(auto& member : members_) {
        if(member.first == key) { member.second = std::move(value); return; }
    }
    members_.emplace_back(key, std::move(value));
}

bool Json::erase(const std::string& key) {
    for(auto it = members_.begin(); it != members_.end(); ++it) {
        if(it->first == key) { members_.erase(it); return true; }
    }
    return false;
}

void Json::sort_keys() {
    if(type_ == Type::Object) {
        for(auto& member : members_) { member.second.sort_keys(); }
        std::stable_sort(members_.begin(), members_.end(),
                         [](const std::pair<std::string, Json>& a, const std::pair<std::string, Json>& b) { return a.first < b.first; });
    } else if(type_ == Type::Array) {
        for(auto& element : elements_) { element.sort_keys(); }
    }
}

std::string format_double(double value, int max_decimal_places) {
    // rapidjson 1.1.0 internal::dtoa + Prettify with maxDecimalPlaces: shortest round trip digits,
    // plain decimal layout for 1e-7 <= |v| < 1e21, TRUNCATED (never rounded) to maxDecimalPlaces.
    if(std::isnan(value) || std::isinf(value)) { return "null"; }
    if(value == 0) { return std::signbit(value) ? "-0.0" : "0.0"; }
    std::string out;
    if(value < 0) { out.push_back('-'); value = -value; }

    char scratch[64];
    auto r = std::to_chars(scratch, scratch + sizeof(scratch), value, std::chars_format::scientific);
    std::string sci(scratch, r.ptr);              // d.ddddde[+-]xx
    size_t epos = sci.find('e');
    std::string digits;
    for(size_t i = 0; i < epos; ++i) { if(sci[i] != '.') { digits.push_back(sci[i]); } }
    int exponent10 = std::atoi(sci.c_str() + epos + 1);
    const int length = static_cast<int>(digits.size());
    const int k = exponent10 - (length - 1);      // value = digits * 10^k
    const int kk = length + k;                    // 10^(kk-1) <= value < 10^kk

    if(0 <= k && kk <= 21) {
        out += digits;
        out.append(static_cast<size_t>(k), '0');
        out += ".0";
    } else if(0 < kk && kk <= 21) {
        std::string body = digits.substr(0, static_cast<size_t>(kk)) + "." + digits.substr(static_cast<size_t>(kk));
        if(0 > k + max_decimal_places) {
            // keep kk + 1 + maxDecimalPlaces characters, then drop trailing zeros but keep one decimal
            int i = kk + max_decimal_places;
            for(; i > kk + 1; --i) { if(body[static_cast<size_t>(i)] != '0') { break; } }
            body.resize(static_cast<size_t>(i > kk + 1 ? i + 1 : kk + 2));
        }
        out += body;
    } else if(-6 < kk && kk <= 0) {
        std::string body = "0." + std::string(static_cast<size_t>(-kk), '0') + digits;
        if(length - kk > max_decimal_places) {
            int i = max_decimal_places + 1;
            for(; i > 2; --i) { if(body[static_cast<size_t>(i)] != '0') { break; } }
            body.resize(static_cast<size_t>(i > 2 ? i + 1 : 3));
        }
        out += body;
    } else if(kk < -max_decimal_places) {
        out += "0.0";
    } else {
        // d[.ddd]e[-]xx
        out.push_back(digits[0]);
        if(length > 1) { out.push_back('.'); out += digits.substr(1); }
        out.push_back('e');
        out += std::to_string(kk - 1);
    }
    return out;
}

void Json::write(std::string& out, int precision, int depth) const {
    switch(type_) {
        case Type::Null: out += "null"; break;
        case Type::Bool: out += bool_ ? "true" : "false"; break;
        case Type::Integer: out += unsigned_ ? std::to_string(static_cast<uint64_t>(int_)) : std::to_string(int_); break;
        case Type::Real: out += format_double(real_, precision); break;
        case Type::String: escape(out, string_); break;
        case Type::Array:
            if(elements_.empty()) { out += "[]"; break; }
            out += "[\n";
            for(size_t i = 0; i < elements_.size(); ++i) {
                out.append(static_cast<size_t>(depth + 1) * 4, ' ');
                elements_[i].write(out, precision, depth + 1);
                if(i + 1 < elements_.size()) { out.push_back(','); }
                out.push_back('\n');
            }
            out.append(static_cast<size_t>(depth) * 4, ' ');
            out.push_back(']');
            break;
        case Type::Object:
            if(members_.empty()) { out += "{}"; break; }
            out += "{\n";
            for(size_t i = 0; i < members_.size(); ++i) {
                out.append(static_cast<size_t>(depth + 1) * 4, ' ');
                escape(out, members_[i].first);
                out += ": ";
                members_[i].second.write(out, precision, depth + 1);
                if(i + 1 < members_.size()) { out.push_back(','); }
                out.push_back('\n');
            }
            out.append(static_cast<size_t>(depth) * 4, ' ');
            out.push_back('}');
            break;
    }
}

std::string Json::dump(int precision) const {
    std::string out;
    write(out, precision, 0);
    return out;
}

}  // namespace phq

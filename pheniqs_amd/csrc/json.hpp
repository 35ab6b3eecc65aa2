// Minimal JSON document model for the compiled job document and the statistics report.
//
// The reference reads its configuration through rapidjson (json.h / json.cpp); only two behaviours of
// that stack matter on the classification path and are reproduced here:
//   * object members keep their document order, because Classifier<T> builds tag_array by walking
//     the "codec" members in order (reference barcode.cpp:75-86, classifier.h:55-60);
//   * the report writer truncates doubles to `float precision` decimals the way
//     PrettyWriter::SetMaxDecimalPlaces does (reference json.cpp:30-46).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace phq {

class Json {
public:
    enum class Type { Null, Bool, Integer, Real, String, Array, Object };
    typedef std::vector<std::pair<std::string, Json>> Members;

    Json() : type_(Type::Null) {}
    static Json boolean(bool v) { Json j; j.type_ = Type::Bool; j.bool_ = v; return j; }
    static Json integer(int64_t v) { Json j; j.type_ = Type::Integer; j.int_ = v; return j; }
    static Json uinteger(uint64_t v) { Json j; j.type_ = Type::Integer; j.int_ = static_cast<int64_t>(v); j.unsigned_ = true; return j; }
    static Json real(double v) { Json j; j.type_ = Type::Real; j.real_ = v; return j; }
    static Json string(const std::string& v) { Json j; j.type_ = Type::String; j.string_ = v; return j; }
    static Json array() { Json j; j.type_ = Type::Array; return j; }
    static Json object() { Json j; j.type_ = Type::Object; return j; }

    // throws ConfigurationError on malformed input
    static Json parse(const std::string& text);

    Type type() const { return type_; }
    bool is_null() const { return type_ == Type::Null; }
    bool is_object() const { return type_ == Type::Object; }
    bool is_array() const { return type_ == Type::Array; }
    bool is_string() const { return type_ == Type::String; }
    bool is_number() const { return type_ == Type::Integer || type_ == Type::Real; }
    bool is_bool() const { return type_ == Type::Bool; }

    const Json* find(const std::string& key) const;
    Json* find(const std::string& key) { return const_cast<Json*>(static_cast<const Json*>(this)->find(key)); }
    const Members& members() const { return members_; }
    Members& members() { return members_; }
    const std::vector<Json>& elements() const { return elements_; }
    std::vector<Json>& elements() { return elements_; }
    bool erase(const std::string& key);
    bool as_bool() const { return bool_; }
    int64_t as_int() const { return type_ == Type::Real ? static_cast<int64_t>(real_) : int_; }
    double as_double() const { return type_ == Type::Real ? real_ : (unsigned_ ? static_cast<double>(static_cast<uint64_t>(int_)) : static_cast<double>(int_)); }
    const std::string& as_string() const { return string_; }

    void push(Json value) { elements_.push_back(std::move(value)); }
    void set(const std::string& key, Json value);

    // keys sorted, 4 space indent, doubles truncated to `precision` decimals: the layout of
    // print_json(report, precision) after sort_json_value (reference json.cpp:39-46,875-893)
    std::string dump(int precision) const;
    void sort_keys();

private:
    Type type_;
    bool bool_ = false;
    bool unsigned_ = false;
    int64_t int_ = 0;
    double real_ = 0;
    std::string string_;
    std::vector<Json> elements_;
    Members members_;
    void write(std::string& out, int precision, int depth) const;
};

// digits of a double as rapidjson's dtoa(value, buffer, maxDecimalPlaces) lays them out
std::string format_double(double value, int max_decimal_places);

}  // namespace phq

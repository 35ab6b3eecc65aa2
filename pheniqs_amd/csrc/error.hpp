// Error values that cross the C ABI.  The numeric space is the reference's process exit codes
// (reference error.h:32-44) so a host that maps them to exit status behaves like `pheniqs` does.
#pragma once

#include <stdexcept>
#include <string>

namespace phq {

enum ErrorCode : int {
    OK = 0,
    UNKNOWN_ERROR = 1,
    INTERNAL_ERROR = 2,
    CONFIGURATION_ERROR = 3,
    OUT_OF_MEMORY_ERROR = 4,
    COMMAND_LINE_ERROR = 5,
    IO_ERROR = 6,
    SEQUENCE_ERROR = 7,
    OVERFLOW_ERROR = 8,
    CORRUPT_AUXILIARY_ERROR = 9,
    JSON_VALIDATION_ERROR = 10,
};

class Error : public std::runtime_error {
public:
    Error(ErrorCode code, const std::string& name, const std::string& message) :
        std::runtime_error(name + " : " + message), code(code), message(message) {}
    const ErrorCode code;
    const std::string message;        // without the error class prefix (error.h:46-76 keeps them apart too)
};
struct InternalError : Error { explicit InternalError(const std::string& m) : Error(INTERNAL_ERROR, "Internal error", m) {} };
struct ConfigurationError : Error { explicit ConfigurationError(const std::string& m) : Error(CONFIGURATION_ERROR, "Configuration error", m) {} };
struct OutOfMemoryError : Error { explicit OutOfMemoryError(const std::string& m) : Error(OUT_OF_MEMORY_ERROR, "Out of memory error", m) {} };
struct SequenceError : Error { explicit SequenceError(const std::string& m) : Error(SEQUENCE_ERROR, "Sequence error", m) {} };
struct OverflowError : Error { explicit OverflowError(const std::string& m) : Error(OVERFLOW_ERROR, "Overflow error", m) {} };
struct IOError : Error { explicit IOError(const std::string& m) : Error(IO_ERROR, "IO error", m) {} };

// the error of class `code` again (a worker thread hands its failure over as a code and a text)
[[noreturn]] inline void throw_error(int code, const std::string& message) {
    switch(code) {
        case CONFIGURATION_ERROR: throw ConfigurationError(message);
        case OUT_OF_MEMORY_ERROR: throw OutOfMemoryError(message);
        case IO_ERROR: throw IOError(message);
        case SEQUENCE_ERROR: throw SequenceError(message);
        case OVERFLOW_ERROR: throw OverflowError(message);
        default: throw InternalError(message);
    }
}

}  // namespace phq

// cusp/exception.h -- same hierarchy and names as the reference (cusp/exception.h:31-80).
#pragma once
#include <exception>
#include <string>

namespace cusp {

class exception : public std::exception {
public:
    exception() {}
    explicit exception(const std::string &msg) : message(msg) {}
    ~exception() noexcept override {}
    const char *what() const noexcept override { return message.c_str(); }
protected:
    std::string message;
};

class not_implemented_exception : public exception { public: explicit not_implemented_exception(const std::string &m) : exception(m) {} };
class io_exception : public exception { public: explicit io_exception(const std::string &m) : exception(m) {} };
class invalid_input_exception : public exception { public: explicit invalid_input_exception(const std::string &m) : exception(m) {} };
class format_exception : public exception { public: explicit format_exception(const std::string &m) : exception(m) {} };
class format_conversion_exception : public format_exception { public: explicit format_conversion_exception(const std::string &m) : format_exception(m) {} };
class runtime_exception : public exception { public: explicit runtime_exception(const std::string &m) : exception(m) {} };

} // namespace cusp

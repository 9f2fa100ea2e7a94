// TEST INFRASTRUCTURE ONLY — see document.h in this directory.
#ifndef ORACLE_RAPIDJSON_SHIM_CURSORSTREAMWRAPPER_H
#define ORACLE_RAPIDJSON_SHIM_CURSORSTREAMWRAPPER_H
#include <string>
namespace rapidjson {
template <typename InputStream> class CursorStreamWrapper {
public:
    explicit CursorStreamWrapper(InputStream &is) : is_(is), line_(1) {}
    void slurp(std::string &out) { is_.slurp(out); }
    void setLine(size_t l) { line_ = l; }
    size_t GetLine() const { return line_; }
private:
    InputStream &is_;
    size_t line_;
};
}  // namespace rapidjson
#endif

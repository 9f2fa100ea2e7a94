// TEST INFRASTRUCTURE ONLY — see document.h in this directory.
#ifndef ORACLE_RAPIDJSON_SHIM_WRITER_H
#define ORACLE_RAPIDJSON_SHIM_WRITER_H
#include <string>
namespace rapidjson {
template <typename OutputStream> class Writer {
public:
    explicit Writer(OutputStream &os) : os_(os) {}
    void emit(const std::string &s) { os_.put(s); }
private:
    OutputStream &os_;
};
}  // namespace rapidjson
#endif

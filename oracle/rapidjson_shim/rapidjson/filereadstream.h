// TEST INFRASTRUCTURE ONLY — see document.h in this directory.
#ifndef ORACLE_RAPIDJSON_SHIM_FILEREADSTREAM_H
#define ORACLE_RAPIDJSON_SHIM_FILEREADSTREAM_H
#include <cstdio>
#include <string>
namespace rapidjson {
class FileReadStream {
public:
    FileReadStream(FILE *fp, char *buffer, size_t bufferSize) : fp_(fp), buf_(buffer), n_(bufferSize), line_(1) {}
    void slurp(std::string &out) {
        size_t got;
        while ((got = fread(buf_, 1, n_, fp_)) > 0) out.append(buf_, got);
    }
    void setLine(size_t l) { line_ = l; }
    size_t line() const { return line_; }
private:
    FILE *fp_;
    char *buf_;
    size_t n_;
    size_t line_;
};
}  // namespace rapidjson
#endif

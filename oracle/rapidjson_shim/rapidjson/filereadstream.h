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
        const long at = ftell(fp_);  // (one allocation of the file's size instead of a string that doubles its way up)
        if (at >= 0 && fseek(fp_, 0, SEEK_END) == 0) {
            const long size = ftell(fp_);
            if (size > at) out.reserve((size_t) (size - at));
            fseek(fp_, at, SEEK_SET);
        }
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

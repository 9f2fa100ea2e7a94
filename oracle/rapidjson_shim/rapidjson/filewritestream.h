// TEST INFRASTRUCTURE ONLY — see document.h in this directory.
#ifndef ORACLE_RAPIDJSON_SHIM_FILEWRITESTREAM_H
#define ORACLE_RAPIDJSON_SHIM_FILEWRITESTREAM_H
#include <cstdio>
#include <string>
namespace rapidjson {
class FileWriteStream {
public:
    FileWriteStream(FILE *fp, char *, size_t) : fp_(fp) {}
    void put(const std::string &s) {
        fwrite(s.data(), 1, s.size(), fp_);
        fflush(fp_);
    }
private:
    FILE *fp_;
};
}  // namespace rapidjson
#endif

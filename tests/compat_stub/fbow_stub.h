// tests/compat_stub/fbow_stub.h -- TEST-ONLY declaration stand-in for Thirdparty/fbow/src/fbow.h: the three types Frame /
// KeyFrame / KeyFrameDatabase hold (fBow, fBow2, Vocabulary) with the members the compat shims use.  The Vocabulary stand-in
// keeps the FILE IMAGE it was read from and hands it out through toStream() (fbow.h:95) -- which is how compat/Frame.cc gets
// the tree into HBM (orbfe_vocab_load) without touching src/System.cc:71-72.  It has NO transform(): the shims must go
// through the C ABI.
#pragma once
#include <cstdint>
#include <fstream>
#include <iterator>
#include <map>
#include <ostream>
#include <string>
#include <vector>

namespace fbow
{
struct _float { // fbow.h:19-27: a float that value-initialises to 0 inside std::map
    float var = 0;
    inline float operator=(float &f) { var = f; return var; }
    inline operator float &() { return var; }
    inline operator float() const { return var; }
};
struct fBow : std::map<uint32_t, _float> {};                   // word id -> weight
struct fBow2 : std::map<uint32_t, std::vector<uint32_t> > {};  // node id -> feature indices

class Vocabulary
{
public:
    void readFromFile(const std::string &path)
    {
        std::ifstream f(path, std::ios::binary);
        image_.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    void fromImage(const void *p, size_t n) { image_.assign((const char *)p, n); } // test helper
    void toStream(std::ostream &str) const { str.write(image_.data(), (std::streamsize)image_.size()); }
    bool isValid() const { return !image_.empty(); }
private:
    std::string image_;
};
} // namespace fbow

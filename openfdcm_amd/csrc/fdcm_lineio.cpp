// fdcm_lineio.cpp -- the reference's line files (.lines / .scene / .tmpl) from C: fdcm_lines_read / _write / _free.
// Format (core/serialization.h:42-57 the packed LinesSerialHeader, :59-97 serializeLines / deserializeLines, :137-139 the
// signature; the container around it is packio v0.2.1, a network dependency that is not in the reference's tree: restated
// from the shipped assets, all 461 of which parse -- SURVEY.md Appendix B):
//     0   16  signature "OPENFDCM" zero padded
//     16   6  3 x u16 container version (0, 2, 0)
//     22   1  u8 compression flag (1 = zlib)
//     23   8  u64 uncompressed body length
//     31   8  u64 compressed body length
//     39   .  zlib stream of: the 45-byte header, then N records of 4 float32 x1 y1 x2 y2 (= the 4 x N column-major LineArray)
// Host code, little-endian like the reference's writer.  Errors are the reference's (std::runtime_error there, a status and
// fdcm_last_error() here).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "fdcm_internal.h"

namespace fdcm {

static const unsigned char kSignature[16] = {'O', 'P', 'E', 'N', 'F', 'D', 'C', 'M', 0, 0, 0, 0, 0, 0, 0, 0};
static constexpr size_t kHeaderBytes = 45;  // sizeof(LinesSerialHeader), packed

template <class T> static void put(std::vector<unsigned char>& b, T v) { unsigned char t[sizeof(T)]; std::memcpy(t, &v, sizeof(T)); b.insert(b.end(), t, t + sizeof(T)); }
template <class T> static T get(const unsigned char* p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }

void lines_read(const char* path, float** out, int64_t* n_out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::string("File '") + path + "' does not exist";  // serialization.h:117-119
    std::vector<unsigned char> blob;
    unsigned char chunk[65536];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) blob.insert(blob.end(), chunk, chunk + got);
    std::fclose(f);
    if (blob.size() < 39 || std::memcmp(blob.data(), kSignature, 16) != 0) throw std::string("File '") + path + "' is not an OPENFDCM line file";
    const bool compressed = blob[22] != 0;
    const uint64_t ulen = get<uint64_t>(&blob[23]), clen = get<uint64_t>(&blob[31]);
    // (a deflate stream expands by at most 1032 : 1: a header that claims more is not worth an allocation)
    if (clen > blob.size() - 39 || ulen > clen * 1032ull + 65536ull) throw std::string("File '") + path + "' is truncated";
    std::vector<unsigned char> body;
    if (compressed) {
        body.resize((size_t)ulen);
        uLongf dlen = (uLongf)ulen;
        if (uncompress(body.data(), &dlen, &blob[39], (uLong)clen) != Z_OK || dlen != ulen) throw std::string("File '") + path + "' is truncated";
    } else {
        body.assign(blob.begin() + 39, blob.begin() + 39 + (size_t)clen);
        if (body.size() != ulen) throw std::string("File '") + path + "' is truncated";
    }
    if (body.size() < kHeaderBytes) throw std::string("File '") + path + "' is truncated";
    // LinesSerialHeader: .. u32 offsetToLineData @30, u8 lineDataFormat @34, u16 lineDataRecordLen @35, u64 lineRecordNum @37
    const uint32_t offset = get<uint32_t>(&body[30]);
    const unsigned char format = body[34];
    const uint16_t record_len = get<uint16_t>(&body[35]);
    const uint64_t n = get<uint64_t>(&body[37]);
    if (format != 0) throw std::string("Line data format not recognized, found <") + std::to_string(record_len) + ">";  // serialization.h:88-91
    // (the reference reads lineRecordNum * lineDataRecordLen bytes into 16-byte lines: only 16 is a well-formed file)
    if (record_len != 16) throw std::string("File '") + path + "': unsupported line record length <" + std::to_string(record_len) + ">";
    if (offset > body.size() || n > (body.size() - offset) / 16) throw std::string("File '") + path + "' is truncated";
    float* lines = (float*)std::malloc(n ? (size_t)n * 16 : 16);
    if (!lines) throw std::string("out of memory");
    std::memcpy(lines, &body[offset], (size_t)n * 16);
    *out = lines;
    *n_out = (int64_t)n;
}

void lines_write(const char* path, const float* lines, int64_t n) {
    if (FILE* probe = std::fopen(path, "rb")) {  // serialization.h:101-105: an existing file is removed first
        std::fclose(probe);
        if (std::remove(path) != 0) throw std::string("File '") + path + "' can't be overwritten";
    }
    std::vector<unsigned char> body;
    body.reserve(kHeaderBytes + (size_t)n * 16);
    const time_t now = time(nullptr);
    tm g{};
    gmtime_r(&now, &g);
    put<uint16_t>(body, 0);                                    // fileSourceID
    put<uint32_t>(body, 0); put<uint16_t>(body, 0); put<uint16_t>(body, 0);  // GUID
    for (int i = 0; i < 8; ++i) body.push_back(0);
    put<uint16_t>(body, 0); put<uint16_t>(body, 10); put<uint16_t>(body, 0);  // version 0.10.0 (openfdcm.cpp:43)
    put<uint16_t>(body, (uint16_t)g.tm_yday); put<uint16_t>(body, (uint16_t)g.tm_year);  // serialization.h:69-70
    put<uint16_t>(body, (uint16_t)kHeaderBytes);               // headerSize
    put<uint32_t>(body, (uint32_t)kHeaderBytes);               // offsetToLineData
    body.push_back(0);                                         // lineDataFormat
    put<uint16_t>(body, 16);                                   // lineDataRecordLen
    put<uint64_t>(body, (uint64_t)n);                          // lineRecordNum
    const unsigned char* src = (const unsigned char*)lines;
    body.insert(body.end(), src, src + (size_t)n * 16);
    uLongf clen = compressBound((uLong)body.size());
    std::vector<unsigned char> comp(clen);
    if (compress(comp.data(), &clen, body.data(), (uLong)body.size()) != Z_OK) throw std::string("Cannot write file '") + path + "'";
    FILE* f = std::fopen(path, "wb");
    if (!f) throw std::string("Cannot write file '") + path + "'";  // serialization.h:107-110
    std::vector<unsigned char> head(kSignature, kSignature + 16);
    put<uint16_t>(head, 0); put<uint16_t>(head, 2); put<uint16_t>(head, 0);
    head.push_back(1);
    put<uint64_t>(head, (uint64_t)body.size()); put<uint64_t>(head, (uint64_t)clen);
    const bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size() && std::fwrite(comp.data(), 1, clen, f) == clen;
    if (std::fclose(f) != 0 || !ok) throw std::string("Cannot write file '") + path + "'";
}

}  // namespace fdcm

// HOST: native OBJ writer, byte-identical to the reference's per-line Python writer
//   save_obj_mesh   /root/reference/lib/mesh_util.py:53-61
//     for v in verts:  'v %.4f %.4f %.4f\n'
//     for f in faces:  f_plus = f + 1;  'f %d %d %d\n' % (f_plus[0], f_plus[2], f_plus[1])     (winding swapped, 1-based)
// Python's '%.4f' and glibc's printf both round the exact binary value correctly, so snprintf gives the same digits.
// Formatting is split over threads (each formats a contiguous chunk into its own buffer), the file is written in order.
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "surs_common.h"

namespace {

void format_verts(const double *v, long long n0, long long n1, std::string &out) {
    out.reserve((size_t)(n1 - n0) * 30);
    char buf[128];
    for (long long i = n0; i < n1; ++i) {
        int k = snprintf(buf, sizeof(buf), "v %.4f %.4f %.4f\n", v[3 * i], v[3 * i + 1], v[3 * i + 2]);
        out.append(buf, (size_t)k);
    }
}

inline char *put_int(char *p, long long x) {
    char tmp[24];
    int n = 0;
    bool neg = x < 0;
    unsigned long long u = neg ? (unsigned long long)(-x) : (unsigned long long)x;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

void format_faces(const int32_t *f, long long n0, long long n1, std::string &out) {
    out.reserve((size_t)(n1 - n0) * 28);
    char buf[96];
    for (long long i = n0; i < n1; ++i) {
        char *p = buf;
        *p++ = 'f'; *p++ = ' ';
        p = put_int(p, (long long)f[3 * i] + 1); *p++ = ' ';
        p = put_int(p, (long long)f[3 * i + 2] + 1); *p++ = ' ';
        p = put_int(p, (long long)f[3 * i + 1] + 1); *p++ = '\n';
        out.append(buf, (size_t)(p - buf));
    }
}

}  // namespace

extern "C" int surs_save_obj_mesh(const char *path, const double *verts, long long n_verts, const int32_t *faces,
                                  long long n_faces, int threads) {
    SURS_REQUIRE(path && (verts || n_verts == 0) && (faces || n_faces == 0) && n_verts >= 0 && n_faces >= 0, "bad argument");
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    if (threads > 64) threads = 64;
    const long long work = n_verts + n_faces;
    if (work < 65536) threads = 1;
    std::vector<std::string> vb(threads), fb(threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            format_verts(verts, n_verts * t / threads, n_verts * (t + 1) / threads, vb[t]);
            format_faces(faces, n_faces * t / threads, n_faces * (t + 1) / threads, fb[t]);
        });
    for (auto &th : pool) th.join();
    FILE *fp = fopen(path, "wb");
    if (!fp) return surs::fail(SURS_E_INVALID, "cannot open %s for writing", path);
    bool ok = true;
    for (int t = 0; t < threads && ok; ++t) ok = fwrite(vb[t].data(), 1, vb[t].size(), fp) == vb[t].size();
    for (int t = 0; t < threads && ok; ++t) ok = fwrite(fb[t].data(), 1, fb[t].size(), fp) == fb[t].size();
    ok = (fclose(fp) == 0) && ok;
    if (!ok) return surs::fail(SURS_E_INVALID, "short write to %s", path);
    return 0;
}

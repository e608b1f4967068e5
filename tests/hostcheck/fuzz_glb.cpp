// fuzz_glb.cpp -- the GLB importer of libwr_hip.so (wr_gltf.cpp + wr_jpeg.cpp, host-only C++)
// built on its own with AddressSanitizer and UndefinedBehaviorSanitizer and fed damaged files.
// The importer reads files from outside (the counterpart of gltf::import_slice and the image
// crate's decoders, src/core/resources.rs:28): whatever the bytes are it must return a status,
// never read or write out of bounds, never throw through the C ABI.
//
//   fuzz_glb <seed.glb> <mutations> [rng seed]
//
// Every mutation of the seed file (byte flips, overwritten 32-bit fields, truncations, spliced
// runs, edits of the JSON text) is parsed, and what parses is walked through every accessor of
// the C ABI.  Exit status 0 when all of them came back; the sanitizers abort otherwise.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wr_abi.h"

// wr_upload_glb's targets live in wr_api.hip (device side): not part of this build.
extern "C" int wr_upload_texture(WrContext*, const uint8_t*, uint32_t, uint32_t, int, uint32_t*) { return WR_ERR_DEVICE; }
extern "C" int wr_upload_material(WrContext*, uint32_t, uint32_t, uint32_t, uint32_t, float, float, uint32_t*) { return WR_ERR_DEVICE; }
extern "C" int wr_upload_mesh(WrContext*, uint32_t, const WrModelVertex*, uint32_t, const uint32_t*, uint32_t, uint32_t) { return WR_ERR_DEVICE; }

static uint64_t rng_state = 0x5EED;
static uint32_t rnd() {  // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}

static unsigned long long walk(const WrGlb* g) {
    unsigned long long sum = 0;
    uint32_t nm = 0, nmat = 0, nt = 0;
    wr_glb_counts(g, &nm, &nmat, &nt);
    for (uint32_t i = 0; i <= nm; ++i) {  // (one past the end on purpose)
        const WrModelVertex* v = nullptr; const uint32_t* idx = nullptr; uint32_t nv = 0, ni = 0, mat = 0;
        if (wr_glb_mesh(g, i, &v, &nv, &idx, &ni, &mat) != WR_OK) continue;
        for (uint32_t k = 0; k < nv; ++k) { uint32_t w; memcpy(&w, &v[k].position[0], 4); sum += w; memcpy(&w, &v[k].bitangent[2], 4); sum += w; }
        for (uint32_t k = 0; k < ni; ++k) { if (idx[k] >= nv) { fprintf(stderr, "index %u of %u vertices accepted\n", idx[k], nv); abort(); } sum += idx[k]; }
        sum += mat;
    }
    for (uint32_t i = 0; i <= nt; ++i) {
        const uint8_t* px = nullptr; uint32_t w = 0, h = 0; int srgb = 0;
        if (wr_glb_texture(g, i, &px, &w, &h, &srgb) != WR_OK) continue;
        for (size_t k = 0; k < (size_t)w * h * 4; k += 97) sum += px[k];
        if (w && h) sum += px[(size_t)w * h * 4 - 1];
    }
    for (uint32_t i = 0; i <= nmat; ++i) {
        uint32_t b = 0, n = 0, m = 0; float me = 0, ro = 0;
        if (wr_glb_material(g, i, &b, &n, &m, &me, &ro) != WR_OK) continue;
        if (b >= nt || n >= nt || m >= nt) { fprintf(stderr, "material names texture %u/%u/%u of %u\n", b, n, m, nt); abort(); }
        sum += b + n + m;
    }
    return sum;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: fuzz_glb <seed.glb> <mutations> [rng seed]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    std::vector<uint8_t> seed;
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) seed.insert(seed.end(), buf, buf + got);
    fclose(f);
    const long rounds = atol(argv[2]);
    if (argc > 3) rng_state = strtoull(argv[3], nullptr, 0) | 1ull;
    uint32_t json_len = 0;
    if (seed.size() >= 20) memcpy(&json_len, &seed[12], 4);
    long parsed = 0;
    unsigned long long sum = 0;
    char err[256];
    for (long r = -1; r < rounds; ++r) {
        std::vector<uint8_t> d = seed;
        if (r >= 0 && !d.empty()) {
            const int edits = 1 + (int)(rnd() % 4);
            for (int e = 0; e < edits; ++e) {
                const uint32_t kind = rnd() % 8;
                const size_t n = d.size();
                if (n == 0) break;
                const bool in_json = json_len && json_len + 20 <= n && (rnd() & 1);
                const size_t pos = in_json ? 20 + rnd() % json_len : rnd() % n;
                switch (kind) {
                    case 0: d[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
                    case 1: d[pos] = (uint8_t)rnd(); break;
                    case 2: { const uint32_t v[6] = {0u, 1u, 0x7fffffffu, 0xffffffffu, 0x80000000u, rnd()}; if (pos + 4 <= n) memcpy(&d[pos], &v[rnd() % 6], 4); break; }
                    case 3: d.resize(pos); break;
                    case 4: { const size_t len = 1 + rnd() % 64, src = rnd() % n; for (size_t k = 0; k < len && pos + k < n && src + k < n; ++k) d[pos + k] = d[src + k]; break; }
                    case 5: { const char* digits = "0123456789-"; d[pos] = (uint8_t)digits[rnd() % 11]; break; }  // numbers of the JSON text
                    case 6: { const char* punct = "{}[],:\"\\"; d[pos] = (uint8_t)punct[rnd() % 8]; break; }
                    default: { const size_t len = 1 + rnd() % 16; d.insert(d.begin() + (long)pos, len, (uint8_t)rnd()); break; }
                }
            }
        }
        WrGlb* g = nullptr;
        err[0] = 0;
        const int rc = wr_glb_parse(d.data(), d.size(), &g, err, sizeof(err));
        if (rc == WR_OK) {
            if (!g) { fprintf(stderr, "WR_OK without a handle\n"); abort(); }
            ++parsed;
            sum += walk(g);
            (void)wr_upload_glb(nullptr, 0, g);  // null context: refused
            wr_glb_free(g);
        } else if (g) {
            fprintf(stderr, "failure %d left a handle\n", rc);
            abort();
        }
        if (r == -1 && rc != WR_OK) { fprintf(stderr, "the seed file itself does not parse: %s\n", err); return 1; }
    }
    printf("%ld mutations, %ld parsed, checksum %llx\n", rounds, parsed, sum);
    return 0;
}

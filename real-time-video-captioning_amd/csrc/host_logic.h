// Host-only logic of libgitcap, free of any HIP dependency so that it also compiles for the CPU under
// -fsanitize=address,undefined (`make asan`, tests/test_host_asan.py): numeric encodings of the weight loader, the
// canonical tensor table, the pipeline's ticket -> slot bookkeeping and the workgroup -> tile map of the residual +
// LayerNorm GEMM (shared with the kernel, which compiles the same function for the device).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gitcap.h"

#if defined(__HIPCC__)
#define GITCAP_HD __host__ __device__
#else
#define GITCAP_HD
#endif

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// an on/off environment switch: set and neither empty nor "0"
inline bool env_flag(const char* name) {
    const char* v = getenv(name);
    return v && v[0] && !(v[0] == '0' && v[1] == 0);
}

inline uint16_t host_f2bf(float f) {    // round-to-nearest-even, NaN stays NaN
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// OCP e4m3fn code of x, or -1 when x is not exactly representable (bias 7, 3 mantissa bits, max 448, no infinities)
inline int host_e4m3_exact(float x) {
    const int sign = std::signbit(x) ? 0x80 : 0;
    const float a = std::fabs(x);
    if (a == 0.f) return sign;
    if (!(a <= 448.f)) return -1;
    int e;
    const float m = std::frexp(a, &e);          // a = m * 2^e, m in [0.5, 1)
    const int E = e - 1 + 7;                    // a = (2m) * 2^(e-1)
    if (E >= 1) {
        const float f = (2.f * m - 1.f) * 8.f;  // mantissa field
        const int M = (int)f;
        if ((float)M != f) return -1;
        return sign | (E << 3) | M;
    }
    const float f = std::ldexp(a, 9);           // subnormal: a = M * 2^-9
    const int M = (int)f;
    if ((float)M != f || M < 1 || M > 7) return -1;
    return sign | M;
}

// value of an e4m3fn code (0x7f / 0xff are NaN)
inline float host_e4m3_value(int code) {
    const int E = (code >> 3) & 15, M = code & 7;
    if (E == 15 && M == 7) return NAN;
    const float a = E ? std::ldexp(1.f + M / 8.f, E - 7) : std::ldexp((float)M, -9);
    return (code & 0x80) ? -a : a;
}

// the power-of-two row scale of e4m3 storage: the smallest 2^e with amax / 2^e <= 448
inline float host_e4m3_row_scale(const float* row, int64_t cols) {
    float amax = 0.f;
    for (int64_t k = 0; k < cols; ++k) amax = std::fmax(amax, std::fabs(row[k]));
    int e = 0;
    if (amax > 0.f) { (void)std::frexp(amax / 448.0f, &e); if (std::ldexp(1.0f, e - 1) * 448.0f >= amax) --e; }
    return std::ldexp(1.0f, e);
}

// Encodes one row (cols values -> q[0..cols), the caller zero-pads up to its row pitch); false when a value is not
// e4m3 x scale exactly (the loader refuses instead of rounding).
inline bool host_e4m3_encode_row(const float* row, int64_t cols, float scale, uint8_t* q) {
    for (int64_t k = 0; k < cols; ++k) {
        const int code = host_e4m3_exact(row[k] / scale);
        if (code < 0) return false;
        q[k] = (uint8_t)code;
    }
    return true;
}

// canonical names -> shape; mirrors gitcap/weights.py:canonical_shapes
inline void expected_shapes(const gitcap_config& c, std::vector<std::pair<std::string, std::vector<int64_t>>>& out) {
    const int64_t Dv = c.enc_width, D = c.dec_width, V = c.vocab_size;
    const int64_t G = c.image_size / c.patch_size, N = G * G + 1, pd = 3LL * c.patch_size * c.patch_size;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { out.emplace_back(n, std::move(s)); };
    add("enc.patch_w", {Dv, pd}); add("enc.cls", {Dv}); add("enc.pos", {N, Dv});
    add("enc.ln_pre.w", {Dv}); add("enc.ln_pre.b", {Dv}); add("enc.ln_post.w", {Dv}); add("enc.ln_post.b", {Dv});
    for (int i = 0; i < c.enc_layers; ++i) {
        const std::string p = "enc.L" + std::to_string(i) + ".";
        add(p + "ln1.w", {Dv}); add(p + "ln1.b", {Dv});
        add(p + "qkv.w", {3 * Dv, Dv}); add(p + "qkv.b", {3 * Dv});
        add(p + "proj.w", {Dv, Dv}); add(p + "proj.b", {Dv});
        add(p + "ln2.w", {Dv}); add(p + "ln2.b", {Dv});
        add(p + "fc1.w", {c.enc_ffn, Dv}); add(p + "fc1.b", {c.enc_ffn});
        add(p + "fc2.w", {Dv, c.enc_ffn}); add(p + "fc2.b", {Dv});
    }
    add("temporal", {c.num_frames > 1 ? c.num_frames : 1, Dv});
    add("vproj.w", {D, Dv}); add("vproj.b", {D}); add("vproj.ln.w", {D}); add("vproj.ln.b", {D});
    add("txt.word", {V, D}); add("txt.pos", {c.max_text_pos, D}); add("txt.ln.w", {D}); add("txt.ln.b", {D});
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = "dec.L" + std::to_string(i) + ".";
        add(p + "qkv.w", {3 * D, D}); add(p + "qkv.b", {3 * D});
        add(p + "ao.w", {D, D}); add(p + "ao.b", {D});
        add(p + "ln1.w", {D}); add(p + "ln1.b", {D});
        add(p + "fc1.w", {c.dec_ffn, D}); add(p + "fc1.b", {c.dec_ffn});
        add(p + "fc2.w", {D, c.dec_ffn}); add(p + "fc2.b", {D});
        add(p + "ln2.w", {D}); add(p + "ln2.b", {D});
    }
    add("head.w", {V, D}); add("head.b", {V});
}

inline bool is_gemm_weight(const std::string& n) {
    if (n == "enc.patch_w" || n == "vproj.w" || n == "head.w") return true;
    auto ends = [&](const char* s) { size_t l = strlen(s); return n.size() >= l && n.compare(n.size() - l, l, s) == 0; };
    return ends("qkv.w") || ends("proj.w") || ends("fc1.w") || ends("fc2.w") || ends("ao.w");
}

// Pipeline bookkeeping (gitcap_greedy_submit / _wait): submission t uses slot t % nslot; a ticket can be waited for
// while its slot has not been handed to a later submission.
inline int ticket_slot(int ticket, int nslot) { return ticket % nslot; }
inline bool ticket_waitable(int ticket, int next_ticket, int nslot) {
    return ticket >= 0 && ticket < next_ticket && ticket >= next_ticket - nslot;
}
// A failed statistics exchange (ExchangeHealth below) leaves every submission made so far with undefined results: the handle keeps
// the ticket count at that moment (`poison_upto`), and a wait on a ticket below it is refused every time it is asked for, so a retry
// cannot hand the ids out (gitcap_greedy_wait / gitcap_beam_search_wait).  Tickets submitted afterwards run on the unfused launches.
inline int poison_mark(int next_ticket) { return next_ticket; }
inline bool ticket_poisoned(int ticket, int poison_upto) { return ticket < poison_upto; }


// ---- residual + LayerNorm GEMM: workgroup -> tile ----------------------------------------------------------------
// The N/256 tiles of a 256-row block wait for each other (statistics exchange, gemm_epilogue.h), so they must be
// consecutive workgroups of ONE XCD's dispatch sequence.  Workgroup b runs on XCD b % 8 as that XCD's (b / 8)-th
// workgroup: XCD x is handed a balanced contiguous range of WHOLE row blocks, their ntn tiles consecutive in its
// sequence.  The grid is padded to 8 * ntn * ceil(nrb / 8); surplus workgroups (the last of every sequence) get no tile.
GITCAP_HD inline int ln_grid_size(int nrb, int ntn) { return 8 * ntn * ((nrb + 7) >> 3); }
// When to use that map: only where it is needed for progress, i.e. on grids of more than 256 tiles, which run in several
// rounds.  A grid that fits the chip is co-resident whatever the map and keeps the plain XCD remap, whose even spread
// (222 tiles: 28 / 27 per XCD) leaves every XCD free CUs for the token-loop kernels of the other streams: handing XCDs
// whole row blocks there (30 / 27) cost the pipelined bench 5 % (1651 vs 1742 captions/s, same box, round 3).
// `cus` = compute units of the device the launch goes to (hipDeviceProp_t::multiProcessorCount, device_cus() in kernels.h:
// 256 on a whole MI355X, fewer on a partitioned one).
inline bool ln_use_rowblock_map(int nrb, int ntn, int cus = 256) { return nrb * ntn > cus; }
GITCAP_HD inline bool ln_tile_of_block(int bid, int nrb, int ntn, int* tm, int* tn) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int c = nrb >> 3, r = nrb & 7;
    const int cnt = c + (xcd < r ? 1 : 0), start = xcd * c + (xcd < r ? xcd : r);
    const int j = slot / ntn;
    if (j >= cnt) return false;
    *tm = start + j; *tn = slot - j * ntn;
    return true;
}

// ---- health of the GEMM + LayerNorm statistics exchange -------------------------------------------------------------------
// A tile of a fused launch waits (bounded) for the sibling tiles of its row block; if the bound is ever hit -- the siblings
// cannot become resident, e.g. a foreign process or a CU-masked stream holds the CUs (co-residency) -- the
// kernel does not trap: it raises a word in host-visible memory and finishes with undefined LayerNorm output.  The host
// reads the word at every entry point (and in gitcap_poll_errors): once raised, the launches issued since the last clean
// check are suspect, the handle switches for good to GEMM + row-kernel launches (same bits, no exchange) and the entry
// point returns GITCAP_ERR_EXCHANGE so that the caller re-runs what it had in flight.
struct ExchangeHealth {
    bool degraded = false;      // fused epilogues switched off for this handle
    int trips = 0;              // times the flag was found raised
};
// flag = the value read from the host-visible word.  Returns true when the caller must reset (sync, clear the word and the
// exchange barriers) and report GITCAP_ERR_EXCHANGE.
inline bool exchange_poll(ExchangeHealth& hs, unsigned flag) {
    if (!flag) return false;
    hs.degraded = true;
    ++hs.trips;
    return true;
}

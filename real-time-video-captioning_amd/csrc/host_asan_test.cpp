// CPU-only driver of the host logic under AddressSanitizer + UBSan (`make asan`): no HIP, no GPU.
#include "host_logic.h"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host_asan_test: %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    // e4m3: every finite code round-trips through value -> exact code
    for (int code = 0; code < 256; ++code) {
        const float v = host_e4m3_value(code);
        if (std::isnan(v)) { CHECK((code & 0x7f) == 0x7f); continue; }
        CHECK(host_e4m3_exact(v) == code);
    }
    CHECK(host_e4m3_exact(448.f) == 0x7e && host_e4m3_exact(449.f) == -1 && host_e4m3_exact(0.3f) == -1);
    CHECK(host_e4m3_exact(std::ldexp(1.f, -9)) == 1 && host_e4m3_exact(std::ldexp(1.f, -10)) == -1);
    CHECK(host_e4m3_exact(INFINITY) == -1 && host_e4m3_exact(NAN) == -1);
    // bf16 rounding: ties to even, NaN preserved
    CHECK(host_f2bf(1.0f) == 0x3f80 && host_f2bf(-2.0f) == 0xc000);
    { uint32_t u = 0x3f808000u; float f; memcpy(&f, &u, 4); CHECK(host_f2bf(f) == 0x3f80); }     // tie -> even (down)
    { uint32_t u = 0x3f818000u; float f; memcpy(&f, &u, 4); CHECK(host_f2bf(f) == 0x3f82); }     // tie -> even (up)
    CHECK((host_f2bf(NAN) & 0x7fc0) == 0x7fc0);
    // row quantisation: values that ARE e4m3 x 2^k encode exactly into a buffer of exactly `cols` bytes; others are refused
    std::mt19937 rng(7);
    for (int trial = 0; trial < 200; ++trial) {
        const int cols = 1 + (int)(rng() % 97), k = (int)(rng() % 21) - 10;
        std::vector<float> row(cols);
        for (auto& v : row) {
            int code;
            do { code = (int)(rng() & 0xff); } while ((code & 0x7f) == 0x7f);
            v = std::ldexp(host_e4m3_value(code), k);
        }
        row[rng() % cols] = std::ldexp(448.f, k);                     // pin the row maximum: the scale must come out as 2^k
        const float scale = host_e4m3_row_scale(row.data(), cols);
        CHECK(scale == std::ldexp(1.f, k));
        std::vector<uint8_t> q(cols);                                 // exact size: an overrun is an ASan report
        CHECK(host_e4m3_encode_row(row.data(), cols, scale, q.data()));
        for (int i = 0; i < cols; ++i) CHECK(host_e4m3_value(q[i]) * scale == row[i]);
        row[0] = 0.3f * scale;
        CHECK(!host_e4m3_encode_row(row.data(), cols, scale, q.data()));
    }
    { std::vector<float> z(5, 0.f); CHECK(host_e4m3_row_scale(z.data(), 5) == 1.f); }
    // canonical tensor table: GIT-base has 12 encoder + 6 decoder layers worth of entries, names unique
    gitcap_config c{};
    c.image_size = 224; c.patch_size = 16; c.enc_width = 768; c.enc_layers = 12; c.enc_heads = 12; c.enc_ffn = 3072;
    c.dec_width = 768; c.dec_layers = 6; c.dec_heads = 12; c.dec_ffn = 3072; c.vocab_size = 30522; c.max_text_pos = 1024; c.num_frames = 6;
    std::vector<std::pair<std::string, std::vector<int64_t>>> shapes;
    expected_shapes(c, shapes);
    std::set<std::string> names;
    int gemm_w = 0;
    for (auto& kv : shapes) { names.insert(kv.first); gemm_w += is_gemm_weight(kv.first); CHECK(!kv.second.empty()); }
    CHECK(names.size() == shapes.size() && shapes.size() == 7 + 12 * 12 + 1 + 4 + 4 + 6 * 12 + 2);
    CHECK(gemm_w == 1 + 12 * 4 + 1 + 6 * 4 + 1);
    CHECK(!is_gemm_weight("enc.L0.qkv.b") && !is_gemm_weight("w") && is_gemm_weight("dec.L5.ao.w"));
    // tickets: four slots, a ticket is waitable until its slot is handed on
    CHECK(ticket_slot(5, 4) == 1 && ticket_waitable(5, 6, 4) && ticket_waitable(2, 6, 4) && !ticket_waitable(1, 6, 4));
    CHECK(!ticket_waitable(6, 6, 4) && !ticket_waitable(-1, 6, 4));
    // a failed exchange with tickets 3, 4, 5 in flight (next_ticket 6): all of them are refused from then on, ticket 6 (submitted
    // afterwards, on the unfused launches) is not; a second failure later moves the mark up
    {
        int upto = 0;
        CHECK(!ticket_poisoned(5, upto));
        upto = poison_mark(6);
        CHECK(ticket_poisoned(3, upto) && ticket_poisoned(5, upto) && !ticket_poisoned(6, upto) && ticket_waitable(5, 7, 4));
        upto = poison_mark(9);
        CHECK(ticket_poisoned(8, upto) && !ticket_poisoned(9, upto));
    }
    // residual + LayerNorm GEMM: every tile exactly once, a row block's tiles = consecutive workgroups of one XCD
    for (int ntn = 1; ntn <= 4; ++ntn)
        for (int nrb = 1; nrb <= 300; ++nrb) {
            const int grid = ln_grid_size(nrb, ntn);
            std::vector<int> seen((size_t)nrb * ntn, -1);
            for (int b = 0; b < grid; ++b) {
                int tm = -1, tn = -1;
                if (!ln_tile_of_block(b, nrb, ntn, &tm, &tn)) continue;
                CHECK(tm >= 0 && tm < nrb && tn >= 0 && tn < ntn && seen[(size_t)tm * ntn + tn] < 0);
                seen[(size_t)tm * ntn + tn] = b;
            }
            for (int tm = 0; tm < nrb; ++tm)
                for (int tn = 0; tn < ntn; ++tn) {
                    CHECK(seen[(size_t)tm * ntn + tn] >= 0);
                    if (tn) CHECK(seen[(size_t)tm * ntn + tn] == seen[(size_t)tm * ntn + tn - 1] + 8);   // same XCD, next in its sequence
                }
        }
    // the row-block map is used as soon as the grid exceeds the device's CUs (also on a partitioned device)
    CHECK(!ln_use_rowblock_map(74, 3) && ln_use_rowblock_map(148, 3));
    CHECK(ln_use_rowblock_map(74, 3, 128) && !ln_use_rowblock_map(40, 3, 128) && ln_use_rowblock_map(11, 3, 32));
    // exchange health: a clean word changes nothing; a raised one degrades the handle for good and is reported every time it is seen
    {
        ExchangeHealth hs;
        CHECK(!exchange_poll(hs, 0u) && !hs.degraded && hs.trips == 0);
        CHECK(exchange_poll(hs, 1u) && hs.degraded && hs.trips == 1);
        CHECK(!exchange_poll(hs, 0u) && hs.degraded && hs.trips == 1);          // the caller cleared the word: quiet again, still degraded
        CHECK(exchange_poll(hs, 7u) && hs.degraded && hs.trips == 2);
    }
    CHECK(pad_to(18912, 256) == 18944 && pad_to(256, 256) == 256 && pad_to(1, 16) == 16);
    std::puts("host_asan_test ok");
    return 0;
}

// Seeded synthetic-cohort generator in the PBWT domain (include/bgt_synth.h). Host code only.
#include "../../include/bgt_synth.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Rng {                       // xoshiro256** seeded by splitmix64 of (seed, row, plane)
    uint64_t s[4];
    static uint64_t mix(uint64_t &x)
    {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t row, uint64_t plane)
    {
        uint64_t x = seed * 0x2545f4914f6cdd1dull + row * 2 + plane + 1;
        for (int i = 0; i < 4; ++i) s[i] = mix(x);
    }
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? (uint64_t)(((unsigned __int128)next() * n) >> 64) : 0; }
    int64_t binomial(int64_t n, double p)          // normal approximation is plenty for a workload model
    {
        const double mean = n * p, sd = std::sqrt(n * p * (1 - p));
        if (mean < 30) {                            // small mean: count Bernoulli arrivals by geometric skips
            int64_t k = 0, at = 0;
            const double lq = std::log1p(-p);
            for (;;) {
                at += 1 + (int64_t)(std::log(1.0 - uniform()) / lq);
                if (at > n) return k;
                ++k;
            }
        }
        const double u1 = uniform(), u2 = uniform();
        const double z = std::sqrt(-2.0 * std::log(1.0 - u1)) * std::cos(6.283185307179586 * u2);
        int64_t k = (int64_t)std::llround(mean + sd * z);
        return std::min<int64_t>(n, std::max<int64_t>(0, k));
    }
};

// canonical byte code of one run (reference pbwt.c:24-36 semantics; written from the format description)
inline void put_run(std::vector<uint8_t> &out, uint32_t len, int bit)
{
    if (len == 0) return;
    if (len < 16) { out.push_back((uint8_t)(len << 1 | bit)); return; }
    for (int d = 7; d >= 0; --d) {
        const uint32_t v = (len >> (4 * d)) & 15u;
        if (v) out.push_back((uint8_t)(((uint32_t)d * 16u + v) << 1 | bit));
    }
}

// random composition of `total` into `parts` positive integers
void compose(Rng &g, int64_t total, int64_t parts, std::vector<int64_t> &out)
{
    out.clear();
    if (parts <= 0) return;
    if (parts == 1) { out.push_back(total); return; }
    std::vector<int64_t> cuts((size_t)parts - 1);
    // distinct cut points in 1..total-1: draw, sort, de-duplicate by nudging (parts << total in practice)
    for (auto &c : cuts) c = 1 + (int64_t)g.below((uint64_t)(total - 1));
    std::sort(cuts.begin(), cuts.end());
    for (size_t i = 1; i < cuts.size(); ++i) if (cuts[i] <= cuts[i - 1]) cuts[i] = cuts[i - 1] + 1;
    // nudging may run past total-1 when parts is close to total: fall back to an even split
    if (!cuts.empty() && cuts.back() > total - 1) {
        for (int64_t i = 0; i < parts; ++i) out.push_back(total / parts + (i < total % parts ? 1 : 0));
        return;
    }
    int64_t prev = 0;
    for (int64_t c : cuts) { out.push_back(c - prev); prev = c; }
    out.push_back(total - prev);
}

// a row of m symbols with `ones` ones arranged in `clusters` runs of ones
void draw_plane(Rng &g, int64_t m, int64_t ones, int64_t clusters, std::vector<uint8_t> &out)
{
    if (ones <= 0) { put_run(out, (uint32_t)m, 0); return; }
    if (ones >= m) { put_run(out, (uint32_t)m, 1); return; }
    const int64_t zeros = m - ones;
    clusters = std::max<int64_t>(1, std::min<int64_t>(clusters, std::min(ones, zeros + 1)));
    std::vector<int64_t> one_runs, zero_runs;
    compose(g, ones, clusters, one_runs);
    // clusters+1 gaps; the two outer ones may be empty: compose (zeros + 2) into clusters+1 positive parts
    // and take 1 off both ends
    if (zeros + 2 >= clusters + 1) {
        compose(g, zeros + 2, clusters + 1, zero_runs);
        zero_runs.front() -= 1; zero_runs.back() -= 1;
    } else {                                       // zeros == clusters-1: all interior gaps are 1
        zero_runs.assign((size_t)clusters + 1, 1);
        zero_runs.front() = 0; zero_runs.back() = 0;
    }
    for (int64_t i = 0; i < clusters; ++i) {
        put_run(out, (uint32_t)zero_runs[i], 0);
        put_run(out, (uint32_t)one_runs[i], 1);
    }
    put_run(out, (uint32_t)zero_runs[clusters], 0);
}

// what is decided per site before any genotype is drawn: stream 2 of (seed, row)
struct SiteDraw { double f; bool mono, multi; int ref, alt; };

SiteDraw draw_site(uint64_t seed, int64_t row)
{
    Rng g(seed, (uint64_t)row, 2);
    SiteDraw s;
    if (g.uniform() < 0.5) s.f = 0.5 / (double)(2 + g.below(200));          // rare half of the spectrum
    else s.f = g.uniform() * 0.5;
    s.mono = g.uniform() < 0.02;                    // a few monomorphic sites so that -f'AC>0' filters
    s.multi = g.uniform() < 0.05;                   // carries <M> (a third allele) in plane 1
    s.ref = (int)g.below(4);
    s.alt = (s.ref + 1 + (int)g.below(3)) & 3;      // a different nucleotide
    return s;
}

void draw_row(int m, uint64_t seed, int64_t row, std::vector<uint8_t> &out, uint32_t len[2])
{
    const SiteDraw sd = draw_site(seed, row);
    {   // plane 0: ALT / <M>
        Rng g(seed, (uint64_t)row, 0);
        const size_t at = out.size();
        const int64_t ones = sd.mono ? 0 : (int64_t)std::llround(sd.f * m);
        const int64_t clusters = 1 + (int64_t)(std::sqrt((double)ones) * (1.0 + g.uniform()));
        draw_plane(g, m, ones, clusters, out);
        len[0] = (uint32_t)(out.size() - at);
    }
    {   // plane 1: missing / <M>
        Rng g(seed, (uint64_t)row, 1);
        const size_t at = out.size();
        // (tuning knob: BGTH_SYNTH_MISSING_PPM overrides the 10^-3 missing rate of SURVEY 8d, e.g. 0 for a fully
        // called panel whose plane 1 is empty except at multi-allelic sites)
        static const double miss = getenv("BGTH_SYNTH_MISSING_PPM") ? atof(getenv("BGTH_SYNTH_MISSING_PPM")) * 1e-6 : 1e-3;
        int64_t ones = miss > 0 ? g.binomial(m, miss) : 0;
        if (sd.multi) ones += g.binomial(m, 2e-2);
        const int64_t clusters = std::max<int64_t>(1, ones - (int64_t)g.below((uint64_t)(ones / 4 + 1)));
        draw_plane(g, m, ones, clusters, out);
        len[1] = (uint32_t)(out.size() - at);
    }
}

}  // namespace

struct bgth_synth_s {
    std::vector<uint8_t> rle;
    std::vector<uint32_t> len;
};

extern "C" bgth_synth_t *bgth_synth_rows(int m, int64_t row0, int64_t n_rows, uint64_t seed, int n_threads)
{
    if (m <= 0 || n_rows < 0) return nullptr;
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n_rows / 1024));
    bgth_synth_t *s = new bgth_synth_s();
    s->len.resize((size_t)n_rows * 2);
    std::vector<std::vector<uint8_t>> part((size_t)n_threads);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            const int64_t a = n_rows * t / n_threads, b = n_rows * (t + 1) / n_threads;
            std::vector<uint8_t> &out = part[(size_t)t];
            out.reserve((size_t)(b - a) * 512);
            for (int64_t r = a; r < b; ++r) draw_row(m, seed, row0 + r, out, &s->len[(size_t)r * 2]);
        });
    }
    for (auto &x : th) x.join();
    size_t total = 0;
    for (auto &p : part) total += p.size();
    s->rle.resize(total);
    size_t at = 0;
    for (auto &p : part) { if (!p.empty()) memcpy(s->rle.data() + at, p.data(), p.size()); at += p.size(); std::vector<uint8_t>().swap(p); }
    return s;
}

extern "C" const uint8_t *bgth_synth_rle(const bgth_synth_t *s) { return s->rle.data(); }
extern "C" const uint32_t *bgth_synth_len(const bgth_synth_t *s) { return s->len.data(); }
extern "C" int64_t bgth_synth_bytes(const bgth_synth_t *s) { return (int64_t)s->rle.size(); }
extern "C" void bgth_synth_free(bgth_synth_t *s) { delete s; }

// Site description of a synthetic row (contig 11, POS = 1000 + 10*row, random distinct
// REF/ALT nucleotides, a third allele on the 5 % multi-allelic sites).
extern "C" void bgth_synth_site(uint64_t seed, int64_t row, int32_t *pos1, char *ref, char *alt, int32_t *n_allele)
{
    const SiteDraw sd = draw_site(seed, row);
    *pos1 = (int32_t)(1000 + 10 * row);
    *ref = "ACGT"[sd.ref]; *alt = "ACGT"[sd.alt];
    *n_allele = sd.multi ? 3 : 2;
}

// CPU-only dump of the host-side table builders (mpc-protocols_amd/csrc/tables.hpp, host_fr.hpp) for
// tests/test_host_tables.py, which compares every value with the big-int oracle.  Built with
// -fsanitize=address,undefined: the table code is the only host arithmetic in the product.
//   usage: host_tables_dump <fr|gl> <n> <d> <t> <id0,id1,...>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../mpc-protocols_amd/csrc/tables.hpp"
#include "../../mpc-protocols_amd/csrc/tables_mfma.hpp"
#include "../../mpc-protocols_amd/csrc/tables_mfma_gl.hpp"

using namespace hbmpc;

template <class H>
static void put(const char* tag, const H& v) {
    uint64_t c[4];
    v.to_canon(c);
    std::printf("%s %016llx%016llx%016llx%016llx\n", tag, (unsigned long long)c[3], (unsigned long long)c[2], (unsigned long long)c[1],
                (unsigned long long)c[0]);
}
template <class H>
static int run(size_t n, size_t d, size_t t, const std::vector<size_t>& ids) {
    const size_t m = d + 1, needed = d + t + 1;
    H w;
    if (!domain_omega(domain_size(n), &w)) return 2;
    put("omega", w);
    const std::vector<H> el = domain_elements<H>(n, n);
    for (const H& e : el) put("alpha", e);
    std::vector<H> xs(m);
    for (size_t i = 0; i < m; ++i) xs[i] = el[ids[i]];
    const auto basis = lagrange_basis(xs);  // basis[i][k]
    for (size_t i = 0; i < m; ++i)
        for (size_t k = 0; k < m; ++k) put("basis", basis[i][k]);
    for (size_t s = m; s < needed; ++s)
        for (size_t i = 0; i < m; ++i) put("verify", horner(basis[i], el[ids[s]]));
    put("inv7", H::from_u64(7).inv());
    put("pow", H::from_u64(3).pow_u64(1000003));
    // the device-constant encodings of one value in every representation it is shipped in
    std::vector<uint32_t> words;
    put_const(words, el[n > 1 ? 1 : 0], std::is_same<H, HGl>::value ? IMPL_GOLD : IMPL_U29);
    std::printf("const");
    for (uint32_t x : words) std::printf(" %08x", x);
    std::printf("\n");
    // the second-chance tables (k_second_chance): window offsets, then every table word in the device-constant form
    const size_t S = ids.size(), rmax = S > needed ? (t < S - needed ? t : S - needed) : 0;
    if (rmax >= 1) {
        const int impl = std::is_same<H, HGl>::value ? IMPL_GOLD : IMPL_U29;
        const SecondTables T = build_second_tables<H>(ids, n, d, needed + rmax, impl);
        SecondTables L;
        L.layout(m, needed + rmax, (size_t)impl_nl(impl));
        std::printf("second %zu %d", needed + rmax, T.n_windows);
        for (int w = 0; w < T.n_windows; ++w) {
            std::printf(" %d", T.win_start[w]);
            if (L.n_windows != T.n_windows || L.win_start[w] != T.win_start[w] || L.ev_off[w] != T.ev_off[w] || L.bc_off[w] != T.bc_off[w]) return 3;
        }
        for (uint32_t x : T.words) std::printf(" %08x", x);
        std::printf("\n");
    }
    // the byte-digit tables of the matrix-core kernels (tables_mfma.hpp / tables_mfma_gl.hpp) over the decode rows of this
    // sender set: built under the sanitizers for every shape they cover, printed for the small ones
    if (m >= 2 && m <= (std::is_same<H, HGl>::value ? MFGL_MAX_M : MF_MAX_M)) {
        const auto rows = recover_coeff_rows<H>(ids, n, d, t);
        std::vector<uint32_t> tab;
        if constexpr (std::is_same<H, HGl>::value) tab = build_mfma_table_gl(rows, m);
        else tab = build_mfma_table(rows, m);
        std::printf("mfma_bytes %zu\n", tab.size() * 4);
        if (tab.size() * 4 <= 65536) {
            std::printf("mfma");
            for (uint32_t x : tab) std::printf(" %08x", x);
            std::printf("\n");
        }
    }
    // the point-pair table of the matrix-core encode (tables_mfma.hpp::build_mfma_bfly_table) for y = X * V on this domain
    if constexpr (std::is_same<H, HFr>::value) {
        if (m >= 2 && m <= MF_BFLY_MAX_M && domain_size(n) >= 4) {
            std::vector<std::vector<HFr>> V(n, std::vector<HFr>(m));
            for (size_t j = 0; j < n; ++j) {
                HFr p = HFr::one();
                for (size_t k = 0; k < m; ++k) V[j][k] = p, p = p * el[j];
            }
            const size_t half = domain_size(n) / 2;
            const auto tab = build_mfma_bfly_table(V, m, half);
            std::printf("bfly_bytes %zu\n", tab.size() * 4);
            if (tab.size() * 4 <= 140000) {
                std::printf("bfly");
                for (uint32_t x : tab) std::printf(" %08x", x);
                std::printf("\n");
            }
        }
    }
    return 0;
}
int main(int argc, char** argv) {
    if (argc < 6) return 1;
    const size_t n = std::strtoul(argv[2], nullptr, 10), d = std::strtoul(argv[3], nullptr, 10), t = std::strtoul(argv[4], nullptr, 10);
    std::vector<size_t> ids;
    for (char* p = std::strtok(argv[5], ","); p; p = std::strtok(nullptr, ",")) ids.push_back(std::strtoul(p, nullptr, 10));
    if (ids.size() < d + t + 1) return 1;
    return std::strcmp(argv[1], "gl") == 0 ? run<HGl>(n, d, t, ids) : run<HFr>(n, d, t, ids);
}

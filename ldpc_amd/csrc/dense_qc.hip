// Shift lists of a dense parity-check matrix, on the host.  The reference hands its decoder the dense
// 0/1 matrix (ldpc.py:225-247; the gym passes its 1022 x 8176 state at every step, ldpc_env.py:370), so
// recognising the block-circulant structure is on the path of every evaluation: 8.4 M entries to look at.
// Here the first row of every block-row gives the shifts, and every other row is checked against them
// (as many non-zeros as the block-row's degree, and a non-zero wherever a shift says) by several threads.
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/ldpc_hip.h"

extern "C" __attribute__((visibility("hidden"))) void ldpc_set_error_(const char *text);

namespace {

template <typename E>
int dense_to_qc_t(const E *H, int m, int n, int Z, int32_t *blk_ptr, int32_t *shifts, int cap)
{
    const int mb = m / Z, nb = n / Z;
    int k = 0;
    for (int b = 0; b < mb; ++b) {
        const E *row = H + (size_t)b * Z * n;
        for (int j = 0; j < nb; ++j) {
            blk_ptr[b * nb + j] = k;
            for (int c = 0; c < Z; ++c)
                if (row[(size_t)j * Z + c] != E(0)) {
                    if (k >= cap) { ldpc_set_error_("more shifts than the caller's array holds"); return LDPC_E_ARG; }
                    shifts[k++] = c;
                }
        }
    }
    blk_ptr[mb * nb] = k;
    std::atomic<bool> bad(false);
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = (size_t)m * n < (1u << 20) ? 1 : (int)std::min<unsigned>(8, hw ? hw : 1);
    auto work = [&](int t) {
        for (int r = t; r < m && !bad.load(std::memory_order_relaxed); r += T) {
            const int b = r / Z, i = r - b * Z;
            if (i == 0) continue;
            const E *row = H + (size_t)r * n;
            int cnt = 0;
            for (int c = 0; c < n; ++c) cnt += row[c] != E(0);
            bool ok = cnt == blk_ptr[(b + 1) * nb] - blk_ptr[b * nb];
            for (int j = 0; j < nb && ok; ++j)
                for (int q = blk_ptr[b * nb + j]; q < blk_ptr[b * nb + j + 1]; ++q) {
                    int c = i + shifts[q];
                    c = c >= Z ? c - Z : c;
                    if (row[(size_t)j * Z + c] == E(0)) { ok = false; break; }
                }
            if (!ok) bad.store(true, std::memory_order_relaxed);
        }
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    return bad.load() ? 1 : 0;
}

}  // namespace

extern "C" int ldpc_dense_to_qc(const void *H, int elem_bytes, int m, int n, int Z, int32_t *blk_ptr,
                                int32_t *shifts, int shifts_cap)
{
    if (!H || !blk_ptr || (!shifts && shifts_cap > 0) || m <= 0 || n <= 0 || Z <= 0 || shifts_cap < 0) {
        ldpc_set_error_("bad arguments");
        return LDPC_E_ARG;
    }
    if (m % Z || n % Z) return 1;
    switch (elem_bytes) {
    case 1: return dense_to_qc_t(static_cast<const uint8_t *>(H), m, n, Z, blk_ptr, shifts, shifts_cap);
    case 4: return dense_to_qc_t(static_cast<const uint32_t *>(H), m, n, Z, blk_ptr, shifts, shifts_cap);
    case 8: return dense_to_qc_t(static_cast<const uint64_t *>(H), m, n, Z, blk_ptr, shifts, shifts_cap);
    default:
        ldpc_set_error_("elem_bytes must be 1, 4 or 8");
        return LDPC_E_ARG;
    }
}

// nbr_dev.h — device bodies of the neighbor-list kernels (nbr.hip), shared with the chain-resident minimiser (chain_min.hip): the same
// code path per centre whichever kernel runs it, so the rows -- slot order, padding, image shifts -- are identical bit for bit.
#ifndef VSSR_NBR_DEV_H
#define VSSR_NBR_DEV_H
#include "vssr_internal.h"

namespace vssr {

__device__ inline int pack_shift(int s0, int s1, int s2) {
    return ((s0 + 128) & 255) | (((s1 + 128) & 255) << 8) | (((s2 + 128) & 255) << 16);
}
__device__ inline void unpack_shift(int p, int &s0, int &s1, int &s2) {
    s0 = (p & 255) - 128;
    s1 = ((p >> 8) & 255) - 128;
    s2 = ((p >> 16) & 255) - 128;
}

// wrap positions into the cell along periodic axes (fractional floor), like the oracle
__device__ __forceinline__ void wrap_atom(int i, const double *__restrict__ pos, const int *__restrict__ atom_cfg,
                                          const double *__restrict__ cell, const double *__restrict__ invcell,
                                          const uint8_t *__restrict__ pbc, double *__restrict__ wpos, int *__restrict__ wrap) {
    int c = atom_cfg[i];
    const double *C = cell + 9 * c, *I = invcell + 9 * c;
    double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    double q[3] = {p[0], p[1], p[2]};
    for (int a = 0; a < 3; ++a) {
        int w = 0;
        if (pbc[3 * c + a]) {
            double f = I[3 * a] * p[0] + I[3 * a + 1] * p[1] + I[3 * a + 2] * p[2];
            w = (int)floor(f);
            q[0] -= w * C[3 * a];
            q[1] -= w * C[3 * a + 1];
            q[2] -= w * C[3 * a + 2];
        }
        wrap[3 * i + a] = w;
    }
    wpos[3 * i] = q[0];
    wpos[3 * i + 1] = q[1];
    wpos[3 * i + 2] = q[2];
}

// LPC lanes per centre atom -- a 16-lane row of a wave (the default: four centres per wave), half a wave or a whole wave:
// lane L tests candidate atom j = a0 + LPC*chunk + L against all periodic images.
// FILL = false counts, FILL = true writes the slots.  Slot order inside a centre is (j ascending, image shift
// lexicographic), reproduced exactly by an exclusive wave scan of the per-lane hit counts, so the CSR is
// identical however the work is spread over lanes.
template <int LPC>
__device__ __forceinline__ int group_excl_scan(int v, int &total) {
    if constexpr (LPC == 64) {
        const int x = wave_incl_scan_i32(v);   // DPP network (vssr_internal.h): the neighbor kernels run 5 of these per centre and are latency-bound
        total = __builtin_amdgcn_readlane(x, 63);
        return x - v;
    } else {   // 16 lanes per centre = one DPP row: the first four steps of the same network (32 lanes: five); the group's last lane holds the total
        static_assert(LPC == 16 || LPC == 32, "a wave, half a wave or one 16-lane row per centre");
        int x = v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);    // row_shr:1
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);    // row_shr:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);    // row_shr:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);    // row_shr:8
        if constexpr (LPC == 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
        total = __shfl(x, (threadIdx.x & 63) | (LPC - 1));
        return x - v;
    }
}

// One centre i, searched by the LPC lanes of its group (lane = threadIdx.x & (LPC - 1); every lane of the group calls this with the
// same i).  row_start: indexed [i] (FILL).
template <bool FILL, int LPC>
__device__ __forceinline__ void nbr_row(int i, const double *__restrict__ wpos, const int *__restrict__ atom_cfg,
      const int *__restrict__ cfg_start, const double *__restrict__ cell, const double *__restrict__ invcell,
      const int *__restrict__ nimg, double rc2, int *__restrict__ deg, const int *__restrict__ row_start, float4 *__restrict__ edge,
      int *__restrict__ edge_S, long long slot_cap, unsigned long long *__restrict__ hits_buf, int hits_stride,
      const unsigned char *__restrict__ active) {
    const int lane = threadIdx.x & (LPC - 1);
    const int c = atom_cfg[i];
    if (active && !active[c]) {   // chain switched off by the relaxation driver: an empty row (its 8 pad slots are never read)
        if (!FILL && lane == 0) deg[i] = 0;
        return;
    }
    const int a0 = cfg_start[c], a1 = cfg_start[c + 1];
    const double *C = cell + 9 * c;
    const int n0 = nimg[3 * c], n1 = nimg[3 * c + 1], n2 = nimg[3 * c + 2];
    const double C0 = C[0], C1 = C[1], C2 = C[2], C3 = C[3], C4 = C[4], C5 = C[5], C6 = C[6], C7 = C[7], C8 = C[8];
    const double px = wpos[3 * i], py = wpos[3 * i + 1], pz = wpos[3 * i + 2];
    // Image pruning for the search: along periodic axis k the separation is at least |df_k + s_k| h_k (h_k = distance between
    // the cell faces = 1 / |row k of the inverse cell|), so only shifts with |df_k + s_k| <= rc / h_k can be inside the cutoff.
    // The bound is widened by 1e-9 (the decision itself stays the exact fp64 test below): typically 0-2 of the 27 images
    // survive.  Non-periodic axes (n_k = 0) keep their single shift 0.
    const double *I = invcell + 9 * c;
    const double I0 = I[0], I1 = I[1], I2 = I[2], I3 = I[3], I4 = I[4], I5 = I[5], I6 = I[6], I7 = I[7], I8 = I[8];
    const double rcut = sqrt(rc2);
    const double wd0 = rcut * sqrt(I0 * I0 + I1 * I1 + I2 * I2) * (1.0 + 1e-9) + 1e-9;
    const double wd1 = rcut * sqrt(I3 * I3 + I4 * I4 + I5 * I5) * (1.0 + 1e-9) + 1e-9;
    const double wd2 = rcut * sqrt(I6 * I6 + I7 * I7 + I8 * I8) * (1.0 + 1e-9) + 1e-9;
    long long base = FILL ? (long long)row_start[i] : 0;
    int run = 0;
    for (int j0 = a0; j0 < a1; j0 += LPC) {
        const int j = j0 + lane;
        const bool have = j < a1;
        double bx = 0, by = 0, bz = 0;
        if (have) { bx = wpos[3 * j] - px; by = wpos[3 * j + 1] - py; bz = wpos[3 * j + 2] - pz; }
        // count the images of j inside the cutoff; remember which ones (bit = running image index) so that the fill
        // pass revisits only the hits instead of all (2 n0 + 1)(2 n1 + 1)(2 n2 + 1) images again
        int cnt = 0;
        unsigned long long hits = 0ull;
        // hits_buf (optional scratch, every configuration scans <= 64 images): the counting pass stores the masks, the
        // fill pass replays them instead of repeating the fp64 search
        unsigned long long *hslot = hits_buf ? hits_buf + (size_t)i * hits_stride + (j - a0) : nullptr;
        if (FILL && hslot) {
            if (have) {
                hits = *hslot;
                cnt = __builtin_popcountll(hits);
            }
        } else if (have) {
            const double f0 = I0 * bx + I1 * by + I2 * bz, f1 = I3 * bx + I4 * by + I5 * bz, f2 = I6 * bx + I7 * by + I8 * bz;
            const int lo0 = n0 ? max(-n0, (int)ceil(-wd0 - f0)) : 0, hi0 = n0 ? min(n0, (int)floor(wd0 - f0)) : 0;
            const int lo1 = n1 ? max(-n1, (int)ceil(-wd1 - f1)) : 0, hi1 = n1 ? min(n1, (int)floor(wd1 - f1)) : 0;
            const int lo2 = n2 ? max(-n2, (int)ceil(-wd2 - f2)) : 0, hi2 = n2 ? min(n2, (int)floor(wd2 - f2)) : 0;
            const int iw1 = 2 * n1 + 1, iw2 = 2 * n2 + 1;
            for (int s0 = lo0; s0 <= hi0; ++s0)
                for (int s1 = lo1; s1 <= hi1; ++s1)
                    for (int s2 = lo2; s2 <= hi2; ++s2) {
                        const int img = ((s0 + n0) * iw1 + (s1 + n1)) * iw2 + (s2 + n2);   // running index of the full scan
                        if (i == j && s0 == 0 && s1 == 0 && s2 == 0) continue;
                        double rx = bx + s0 * C0 + s1 * C3 + s2 * C6;
                        double ry = by + s0 * C1 + s1 * C4 + s2 * C7;
                        double rz = bz + s0 * C2 + s1 * C5 + s2 * C8;
                        double d2 = rx * rx + ry * ry + rz * rz;
                        if (d2 <= rc2 && d2 > 0.0) {
                            ++cnt;
                            if (img < 64) hits |= 1ull << img;
                        }
                    }
            if (!FILL && hslot) *hslot = hits;
        }
        int total;
        const int off = group_excl_scan<LPC>(cnt, total);
        if (FILL && cnt > 0) {
            long long slot = base + run + off;
            const int w1 = 2 * n1 + 1, w2 = 2 * n2 + 1;
            if ((2 * n0 + 1) * w1 * w2 <= 64) {   // the usual case: replay the set bits in ascending (= lexicographic) order
                while (hits) {
                    const int img = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const int s0 = img / (w1 * w2) - n0, s1 = (img / w2) % w1 - n1, s2 = img % w2 - n2;
                    const double rx = bx + s0 * C0 + s1 * C3 + s2 * C6;
                    const double ry = by + s0 * C1 + s1 * C4 + s2 * C7;
                    const double rz = bz + s0 * C2 + s1 * C5 + s2 * C8;
                    if (slot < slot_cap) {
                        edge[slot] = make_float4((float)rx, (float)ry, (float)rz, __int_as_float(j));
                        edge_S[slot] = pack_shift(s0, s1, s2);
                    }
                    ++slot;
                }
            } else {
                for (int s0 = -n0; s0 <= n0; ++s0)
                    for (int s1 = -n1; s1 <= n1; ++s1)
                        for (int s2 = -n2; s2 <= n2; ++s2) {
                            if (i == j && s0 == 0 && s1 == 0 && s2 == 0) continue;
                            double rx = bx + s0 * C0 + s1 * C3 + s2 * C6;
                            double ry = by + s0 * C1 + s1 * C4 + s2 * C7;
                            double rz = bz + s0 * C2 + s1 * C5 + s2 * C8;
                            double d2 = rx * rx + ry * ry + rz * rz;
                            if (d2 > rc2 || d2 <= 0.0) continue;
                            if (slot < slot_cap) {
                                edge[slot] = make_float4((float)rx, (float)ry, (float)rz, __int_as_float(j));
                                edge_S[slot] = pack_shift(s0, s1, s2);
                            }
                            ++slot;
                        }
            }
        }
        run += total;
    }
    if (FILL) {
        const int padded = max((run + 3) & ~3, 8);   // at least two quads per centre (an isolated atom gets 8 pads): the edge
        const long long slot = base + run + lane;    // kernels prefetch two steps ahead across at most one centre boundary
        if (lane < padded - run && slot < slot_cap) {
            edge[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            edge_S[slot] = pack_shift(0, 0, 0);
        }
    } else if (lane == 0) {
        deg[i] = run;
    }
}

// reverse-edge slot: for slot (i -> j, S') find (j -> i, -S') in j's row.  One wave per centre, lane per slot.
// (row_start: indexed [i], [i + 1], [j], [j + 1])
template <int LPC>
__device__ __forceinline__ void rev_row(int i, const int *__restrict__ row_start, const float4 *__restrict__ edge,
                                        const int *__restrict__ edge_S, int *__restrict__ rev) {
    const int lane = threadIdx.x & (LPC - 1);
    for (int e = row_start[i] + lane; e < row_start[i + 1]; e += LPC) {
        int j = __float_as_int(edge[e].w);
        int found = -1;
        if (j >= 0) {
            int s0, s1, s2;
            unpack_shift(edge_S[e], s0, s1, s2);
            const int want = pack_shift(-s0, -s1, -s2);
            // j's row is sorted by neighbor index (pads, index -1, at its end): binary search to the first entry with
            // neighbor i, then walk the (few) periodic images of that pair
            int lo = row_start[j], hi = row_start[j + 1];
            const int row_end = hi;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int jm = __float_as_int(edge[mid].w);
                if (jm >= 0 && jm < i) lo = mid + 1; else hi = mid;
            }
            for (int e2 = lo; e2 < row_end && __float_as_int(edge[e2].w) == i; ++e2)
                if (edge_S[e2] == want) {
                    found = e2;
                    break;
                }
        }
        rev[e] = found;
    }
}

}  // namespace vssr
#endif

// Plain ANSint on arbitrary values (ans_int.hpp:38-98,114-195).
//
// The reference sizes every array of its model by the list's largest value (freqs(max_sym + 1), ans_int.hpp:41-51)
// and its prelude codes max_sym + 1 items (ans_util.hpp:46-63).  Everything that reaches the payload, though, depends
// on the PRESENT symbols only: adjust_freqs sorts the non-zero counts (ties by symbol index), sums entropies over
// them (absent symbols add +0.0, in index order), and a symbol's base is the frequency mass below it.  So a block is
// modelled in RANK space -- its distinct values sorted, value -> 0-based rank (k_pa_remap, mode 1), the same model /
// encoder / decoder kernels as ever on at most 16384 ranks -- and only the prelude lives in value space:
//
//   k_int_sparse_prelude   the reference's prelude for the block: vbyte(max_sym), log2 M, interpolative code of
//                          inc[s] = (frequency mass up to s) + s over ALL s <= max_sym -- without visiting them
//   k_int_sparse_parse     the inverse: present symbols and their frequencies from such a prelude
//   k_int_unmap            rank -> value on the decoded block
//
// The code of interp.hpp:65-79 is a pre-order walk of a balanced tree over the item indices; a node (items
// [a, a + n), bounds from its ancestors) writes bits iff u = high - n2 - low - n1 + 1 > 1, and u - 1 is the
// frequency mass of the CLOSED range [a, a + n] (the item right of the node included; + 1 for nodes ending at the
// last item).  A node without a present symbol in that range is silent, and so is its whole subtree.  Every other
// node is written by its OWNER, the first present symbol in [a, a + n]:
//   * the owner meets the node on its own walk: the nodes that hold it as an item (root -> the node whose middle
//     item it is: phase 1), then the nodes that END just left of it (the left child of that node and its right
//     spine: phase 2);
//   * nodes owned by an earlier symbol precede nodes owned by a later one in pre-order (they start further left,
//     so they are ancestors or lie to the left), and one owner's nodes are nested, i.e. in walk order.
// So the bit offset of a node is (bits of all earlier owners) + (bits of the owner's earlier nodes): two walks per
// present symbol with an exclusive scan in between, no sorting, 2 log2(max_sym) steps per symbol.
#pragma once

#include "ansx_kernels.h"

#define ANSX_SP_MAX_SIGMA 16384u      // distinct values per block (ranks are symbols of the 16384-slot ANSint model)
#define ANSX_SP_VALUE_LIMIT (1u << 30)  // values below 2^30: the code's universe M + max_sym + 2 stays below 2^31

// number of values <= t among vals[lo .. hi) (ascending)
__device__ __forceinline__ u32 sp_count_le(const u32* __restrict__ vals, u32 lo, u32 hi, u32 t)
{
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (vals[mid] <= t) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The nodes symbol j owns, in pre-order: emit(code) for each.  vals: the block's distinct values (ascending),
// tab: the rank-space encoder table (base = frequency mass below the rank), N = max_sym + 1, u = M + N + 1.
template <typename F>
__device__ __forceinline__ void sp_walk(const u32* __restrict__ vals, const ansx_enc_entry* __restrict__ tab, u32 sigma,
    u32 j, u32 N, u32 u, u32 M, F&& emit)
{
    const u32 s = vals[j];
    const long long prev = j ? (long long)vals[j - 1] : -1ll;
    const u32 Cprev = tab[j].base, Cj = Cprev + tab[j].freq;
    auto mass_upto = [&](u32 t) -> u32 {  // frequency mass of the symbols <= t, for t >= s
        // (galloping from j: deep in the tree t is a few symbols to the right of s, and these reads go to L2)
        u32 bound = 1;
        while (j + bound < sigma && vals[j + bound] <= t) bound <<= 1;
        const u32 idx = sp_count_le(vals, j + (bound >> 1), j + bound < sigma ? j + bound : sigma, t);
        return idx < sigma ? tab[idx].base : M;
    };
    u32 a = 0, n = N;
    bool have_cr = false;
    u32 cr = 0;  // mass up to the item right of the node (a + n < N)
    u32 h, m;
    // phase 1: the nodes that hold s as an item
    for (;;) {
        h = (n + 1) >> 1;
        m = a + h - 1;
        const bool owned = prev < (long long)a;
        u32 cm = 0;
        if (owned) {
            if (!have_cr && a + n < N) {
                cr = mass_upto(a + n);
                have_cr = true;
            }
            cm = m < s ? Cprev : mass_upto(m);
            emit(interp_code<u32>(N, u, a, n, 0u, a ? Cprev + a - 1u : 0u, a + n < N ? cr + a + n : 0u, cm + m));
        }
        if (m == s) break;
        if (s < m) {  // left: the node now ends left of item m
            n = h - 1;
            cr = cm;
            have_cr = owned;
        } else {
            a = m + 1;
            n = n - h;
        }
    }
    // phase 2: the nodes that end just left of s
    n = h - 1;
    while (n > 0) {
        const u32 h2 = (n + 1) >> 1, m2 = a + h2 - 1;
        if (prev < (long long)a) emit(interp_code<u32>(N, u, a, n, 0u, a ? Cprev + a - 1u : 0u, Cj + s, Cprev + m2));
        a = m2 + 1;
        n = n - h2;
    }
}

// One workgroup of 256 threads per block.  Dynamic LDS: off[sigma_cap] + bits[bits_cap + 2] words, sized by the host from the
// call's most distinct values per block (read back with the model's other maxima) so that several workgroups share a CU -- a
// workgroup is a chain of dependent L2 reads; bits_cap = 2 words per distinct value (+ 64): a block whose code is longer raises the
// violation flag and the host repeats the call with the full 16384 + 16384 words.
// limit_bytes: room for vbyte + log2 M + code in the block's scratch slot (the model's 4 x 16384 bytes); a prelude
// that does not fit fails the call with ANSX_ERR_DOMAIN (it would take ~32 bits per distinct value).
__global__ __launch_bounds__(256) void k_int_sparse_prelude(ansx_geo g, u32 NSP, const u32* __restrict__ alpha,
    const ansx_enc_entry* __restrict__ table, ansx_blk* __restrict__ blk, u8* __restrict__ scratch, u64 scr_stride,
    u32 sigma_cap, u32 bits_cap, u32 limit_bytes, u32* __restrict__ gflags)
{
    extern __shared__ u32 sp_lds[];
    __shared__ u32 sh_part[8];
    const u32 tid = threadIdx.x, b = blockIdx.x;
    ansx_blk* B = &blk[b];
    if (B->status || !B->resolved) {
        if (tid == 0) B->prelude_bytes = 0;
        return;
    }
    u32* off = sp_lds;
    u32* bits = sp_lds + sigma_cap;
    const u32 sigma = B->sp_sigma;
    if (sigma > sigma_cap) {  // (cannot happen: the cap is the call's maximum)
        if (tid == 0) {
            B->prelude_bytes = 0;
            B->status = 1;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
        }
        return;
    }
    const u32 logM = B->logM, M = 1u << logM;
    const u32* vals = alpha + (u64)b * g.block_ints;
    const ansx_enc_entry* tab = table + (u64)b * NSP;
    const u32 N = vals[sigma - 1] + 1u;
    const u32 u = M + N + 1u;
    for (u32 j = tid; j < sigma; j += 256) {
        u32 nb = 0;
        sp_walk(vals, tab, sigma, j, N, u, M, [&](const ansx_code& c) { nb += c.len; });
        off[j] = nb;
    }
    __syncthreads();
    const u32 per = (sigma + 255) / 256;
    const u32 lo = tid * per, hi = (lo + per) < sigma ? (lo + per) : sigma;
    u32 sum = 0;
    for (u32 j = lo; j < hi; j++) sum += off[j];
    u32 total_bits;
    u32 run = block_excl_scan<u32>(sum, sh_part, tid, 256, &total_bits);
    for (u32 j = lo; j < hi; j++) {
        const u32 t = off[j];
        off[j] = run;
        run += t;
    }
    const u32 nwords = (total_bits + 31) >> 5;
    const u32 ms = N - 1;
    u32 vb = 1;
    for (u32 t = ms; t >= 128; t >>= 7) vb++;
    const u32 p = vb + 1;
    if (p + nwords * 4 > limit_bytes || nwords > bits_cap) {  // (workgroup-uniform)
        if (tid == 0) {
            B->prelude_bytes = 0;
            B->status = 1;  // the encoder skips the block
            atomicOr(&gflags[ANSX_G_ERR], p + nwords * 4 > limit_bytes ? 1u << 6 /* ANSX_ERR_DOMAIN */ : 1u << ANSX_G_VIOL_BIT);
        }
        return;
    }
    for (u32 w = tid; w <= nwords; w += 256) bits[w] = 0;
    __syncthreads();
    for (u32 j = tid; j < sigma; j += 256) {
        u32 o = off[j];
        sp_walk(vals, tab, sigma, j, N, u, M, [&](const ansx_code& c) {
            if (c.len) {
                const u32 w = o >> 5, sh = o & 31;
                atomicOr(&bits[w], c.code << sh);
                if (sh + c.len > 32) atomicOr(&bits[w + 1], c.code >> (32 - sh));
                o += c.len;
            }
        });
    }
    __syncthreads();
    u8* out = scratch + (u64)b * scr_stride;
    if (tid == 0) {  // vbyte(max_sym) (vbyte.hpp:57-80) + log2(M) byte (ans_util.hpp:49-51)
        u32 t = ms, q = 0;
        while (t >= 128) {
            out[q++] = (u8)((t & 127) | 128);
            t >>= 7;
        }
        out[q++] = (u8)(t & 127);
        out[q] = (u8)logM;
        B->hdr_bytes = 0;
        B->prelude_bytes = p + nwords * 4;
    }
    for (u32 w = tid; w < nwords; w += 256) st_u32_unaligned(out + p + 4 * w, bits[w]);
}

// Decoder side, one lane per block: the present symbols of a prelude and their frequencies, from a walk over the
// nodes that carry bits.  Decoding is pre-order (the order of the bits), the items come out in index order: a node is
// decoded when it is first reached and pushed; it is popped -- its item reported -- when its left subtree is done.
// Two consecutive decoded items t1 < t2 have only absent symbols between them (a present symbol is the middle item of
// a node that carries bits), so nfreq[t2] = inc[t2] - inc[t1] - (t2 - t1).
// Output: alpha[b][r] = value of rank r, g_cum row in rank space (gc[r + 1] = mass up to rank r, + r),
// binfo[b] = {sigma, log2 M, 0, error}, sp_info[b] = {sigma, 0, error, 0} (the layout k_pa_unmap's callers use).
__global__ __launch_bounds__(64) void k_int_sparse_parse(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, u64 payload_off, u32 maxM, u32* __restrict__ g_cum, uint4* __restrict__ binfo,
    u32* __restrict__ alpha, uint4* __restrict__ sp_info, u32* __restrict__ gflags)
{
    __shared__ uint4 stk[32][64];
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x * 64 + lane;
    if (b >= g.nblocks) return;
    const parse_hdr H = parse_header<false>(cont, g, 0xFFFFFFFFu, block_off, payload_off, 0xFFFFFFFFu, maxM, b, nullptr);
    u32 err = H.err;
    const u32 N = H.ns, logM = H.logM;
    if (!err && (N == 0 || N > ANSX_SP_VALUE_LIMIT || logM > 27)) err = 1;
    u32 sigma = 0;
    if (!err) {
        const u32 M = 1u << logM;
        const u32 cap = g.block_ints < ANSX_SP_MAX_SIGMA ? g.block_ints : ANSX_SP_MAX_SIGMA;  // a block has at most block_ints distinct values
        const u8* bp = H.stream + H.pos;
        const u32 avail = H.sbytes - H.pos;
        u32* cum = g_cum + (u64)b * (NSP + 8);
        u32* al = alpha + (u64)b * g.block_ints;
        u64 w0 = ld_u64_unaligned(bp), w1 = ld_u64_unaligned(bp + 8);  // (a stream has at least 38 bytes: parse_header)
        u32 consumed = 0, next_byte = 16, total_bits = 0;
        const u32 maxbits = avail * 8;
        auto getbits = [&](u32 nbits) -> u32 {
            if (nbits == 0) return 0u;
            u64 v = w0 >> consumed;
            if (consumed + nbits > 64) v |= w1 << (64 - consumed);
            consumed += nbits;
            total_bits += nbits;
            if (consumed >= 64) {
                consumed -= 64;
                w0 = w1;
                const u32 nb_ = next_byte + 8 <= avail ? next_byte : avail - 8;  // never past the block's own bytes
                w1 = ld_u64_unaligned(bp + nb_);
                next_byte += 8;
            }
            return (u32)(v & ((nbits >= 32) ? 0xFFFFFFFFull : ((1ull << nbits) - 1ull)));
        };
        const u32 u = M + N + 1u;
        u32 sp = 0;
        u32 a = 0, n = N, low = 1, high = u + 1;
        long long tprev = -1, incprev = -1;  // inc[-1] = -1: inc[t] - t is the mass up to t
        u32 mass = 0;
        // A single flat loop (every lane walks another block, so both halves of the body run for the wave anyway): (1) if the
        // current node is empty or silent, report the item on top of the stack and move to its right subtree; (2) decode the
        // current node, if it carries bits, and go left.  One pop and one decode per iteration: as many iterations as nodes.
        auto shape = [&](u32& h, u32& n1, u32& n2, u32& U) -> bool {  // node (n != 0): its split and code range; false = malformed
            h = (n + 1) >> 1, n1 = h - 1, n2 = n - h;
            if (high < n2 + low + n1 || total_bits > maxbits || sp >= 32) return false;
            U = high - n2 - low - n1 + 1;
            return U <= u + 1;
        };
        for (;;) {
            u32 U = 1, h = 0, n1 = 0, n2 = 0;
            if (n != 0 && !shape(h, n1, n2, U)) {
                err = 1;
                break;
            }
            if (n == 0 || U == 1) {  // empty, or silent (and so is everything below)
                if (sp == 0) break;
                sp--;
                const uint4 e = stk[sp][lane];
                const long long t = e.x, inc = (long long)e.y - 1;
                const long long nf = inc - incprev - (t - tprev);
                if (nf < 0 || nf > (long long)M || (nf > 0 && sigma >= cap)) {
                    err = 1;
                    break;
                }
                if (nf > 0) {
                    mass += (u32)nf;
                    al[sigma] = (u32)t;
                    cum[sigma + 1] = mass + sigma;
                    sigma++;
                }
                tprev = t;
                incprev = inc;
                a = e.x + 1;  // right subtree
                n = e.z;
                low = e.y + 1;
                high = e.w;
                U = 1;
                if (n != 0 && !shape(h, n1, n2, U)) {
                    err = 1;
                    break;
                }
            }
            if (n != 0 && U != 1) {
                const u32 bb = 32 - __clz(U - 1);  // read_center_mid (interp.hpp:47-63)
                const u32 mth = (u32)((1ull << bb) - U);
                const u32 dh = U - (1u << (bb - 1));
                u32 val = getbits(bb - 1) + 1;
                if (val > mth) val = (2 * val + getbits(1)) - mth - 1;
                val += dh;
                if (val > U) val -= U;
                const u32 v = low + n1 - 1 + val;
                stk[sp][lane] = make_uint4(a + h - 1, v, n2, high);
                sp++;
                n = n1;
                high = v - 1;
            } else if (n != 0) {
                n = 0;  // silent right subtree: the next iteration pops again
            }
        }
        if (!err && (mass != M || sigma == 0 || (long long)(N - 1) != tprev)) err = 1;  // the last item (max_sym) is present
    }
    binfo[b] = make_uint4(sigma ? sigma : 1u, logM, 0u, err);
    sp_info[b] = make_uint4(sigma, 0u, err, 0u);
    if (err) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
}

// rank -> value on the decoded block (0-based ranks)
__global__ __launch_bounds__(256) void k_int_unmap(ansx_geo g, const u32* __restrict__ alpha, const uint4* __restrict__ sp_info,
    u32* __restrict__ out, u32* __restrict__ gflags)
{
    const u32 b = blockIdx.x;
    const uint4 pi = sp_info[b];
    if (pi.z) return;
    const u32 nb = geo_block_n(g, b);
    const u32* al = alpha + (u64)b * g.block_ints;
    u32* o = out + (u64)b * g.block_ints;
    const u32 sigma = pi.x;
    u32 bad = 0;
    for (u32 i0 = threadIdx.x; i0 < nb; i0 += 256 * 8) {
        u32 r[8], x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = i0 + 256 * k < nb ? o[i0 + 256 * k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            bad |= r[k] < sigma ? 0u : 1u;
            x[k] = al[r[k] < sigma ? r[k] : 0u];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (i0 + 256 * k < nb && r[k] < sigma) o[i0 + 256 * k] = x[k];
    }
    if (bad) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
}

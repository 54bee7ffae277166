// k_wire.hip — wire-format fan-out buffers on the device.
//
// What the reference does per fan-out message, three marshals deep:
//   data.go:293-318        ChannelDataUpdateMessage{Data: anypb.New(update)} for (conn, channel)
//   connection.go:57-83    MessagePack{ChannelId, MsgType = CHANNEL_DATA_UPDATE, MsgBody}; a pack whose
//                          size is >= MaxPacketSize - PacketHeaderSize is dropped
//   connection.go:626-714  flush: packs are appended to a Packet while proto.Size(packet) <= 65535,
//                          the pack that would overflow opens the next packet; every packet goes out
//                          behind the 5-byte tag {'C','H', size_hi, size_lo, compression}
// Here the host supplies, per channel, the serialized google.protobuf.Any of its (merged) update
// and of its full state; everything else — the three nested length-delimited headers, the greedy
// packet split and the tags — is produced per connection from the tick's fan-out records:
//
//   k_wire_layout  one wave per connection walks its records in stream order (subscription
//                  segments, records inside them): entry size of every record from the channel id's
//                  varint length and the payload length (wave prefix sum), greedy packet cuts with
//                  wave-uniform bookkeeping; per record its byte offset in the connection's stream
//                  and, for the first record of a packet, that packet's length (for the tag)
//   scan           connection stream lengths -> bases in the wire arena
//   k_wire_copy    one lane per record: tag (if first of its packet) + headers are composed in
//                  registers and streamed with the payload through a byte FIFO into the arena
//                  (dword stores once the destination is aligned)
// Bytes written per message = 5/packet + ~12 header + payload: this is the P*M term of SURVEY §8d.
#include "chd_kernels.h"

#define WIRE_MAX_PACKET 65535u
#define WIRE_DROP_SIZE 65530u  // MaxPacketSize - PacketHeaderSize (connection.go:72)

__device__ __forceinline__ uint32_t vlen(uint32_t v) { return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u; }

struct WireMsg {
    uint32_t chan, any_len, body_len, mp_len, entry;  // entry = bytes inside the Packet (0: dropped by Send)
    const uint8_t *pay;
};

__device__ __forceinline__ WireMsg wire_msg(const WorldDev &w, const WireDev &x, chd_fanout_rec rec, uint32_t pos) {
    WireMsg m;
    const uint32_t full = rec.conn >> 31;
    m.chan = rec.channel;
    if (pos & CHD_POS_CELL) {
        const uint32_t c = pos & ~CHD_POS_CELL;
        m.any_len = x.len_cell[full][c];
        m.pay = x.pay_cell[full] + (size_t)c * x.stride[full];
    } else {
        const uint32_t slot = w.ce_slot[pos];
        m.any_len = x.len_ent[full][slot];
        m.pay = x.pay_ent[full] + (size_t)slot * x.stride[full];
    }
    m.body_len = 1u + vlen(m.any_len) + m.any_len;                       // ChannelDataUpdateMessage.data = 1
    m.mp_len = (m.chan ? 1u + vlen(m.chan) : 0u) + 2u                    // channelId = 1 (omitted if 0), msgType = 4 -> 0x20 0x08
               + 1u + vlen(m.body_len) + m.body_len;                     // msgBody = 5
    m.entry = m.mp_len >= WIRE_DROP_SIZE ? 0u : 1u + vlen(m.mp_len) + m.mp_len;  // Packet.messages = 1
    return m;
}

// One wave per connection slot.
__global__ void __launch_bounds__(256) k_wire_layout(WorldDev w, WireDev x) {
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t pkt_base = 0;   // stream offset of the current packet's tag
    uint32_t pkt_used = 0;   // bytes of entries in the current packet
    uint64_t pkt_first = ~0ull;  // record index of the current packet's first entry
    uint32_t npk = 0, ndropped = 0;
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        const size_t pbase = (size_t)s * w.capq;
        const uint64_t rbase = w.rec_ub[s];
        for (uint32_t p = 0; p < cnt; p++) {
            const uint32_t n = w.pair_nrec[pbase + p];
            const uint64_t seg = rbase + w.pair_rel[pbase + p];
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool valid = i < n;
                uint32_t entry = 0;
                if (valid) entry = wire_msg(w, x, w.recs[seg + i], w.rec_pos[seg + i]).entry;
                if (valid && entry == 0) x.rec_woff[seg + i] = 0xFFFFFFFFu;  // dropped
                ndropped += (uint32_t)__popcll(__ballot(valid && entry == 0));
                // inclusive prefix of the entry sizes over the chunk
                uint32_t cum = entry;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(cum, d);
                    if ((int)lane >= d) cum += o;
                }
                uint32_t start_lane = 0, cum_before_start = 0;
                for (;;) {
                    // bytes the current packet would hold after this lane's entry
                    const uint32_t rel = pkt_used + (cum - cum_before_start);
                    const bool mine = valid && lane >= start_lane;
                    const uint64_t over = __ballot(mine && entry != 0 && rel > WIRE_MAX_PACKET);
                    const uint32_t f = over ? (uint32_t)__ffsll((unsigned long long)over) - 1u : 64u;
                    // lanes [start_lane, f) stay in the current packet
                    if (mine && lane < f && entry != 0) {
                        x.rec_woff[seg + i] = (uint32_t)(pkt_base + 5u + rel - entry);
                        x.rec_wtag[seg + i] = 0;
                    }
                    // the first entry of the current packet, if it is in this range
                    const uint64_t firsts = __ballot(mine && lane < f && entry != 0);
                    if (pkt_first == ~0ull && firsts) pkt_first = seg + i0 + (uint32_t)__ffsll((unsigned long long)firsts) - 1u;
                    const uint32_t cum_f = f < 64 ? __shfl(cum - entry, (int)f) : __shfl(cum, 63);  // prefix before lane f
                    pkt_used += cum_f - cum_before_start;
                    if (f == 64) break;
                    // close the packet (flush, connection.go:646-661): lane f's entry opens the next one
                    if (lane == 0 && pkt_first != ~0ull) x.rec_wtag[pkt_first] = 0x80000000u | pkt_used;
                    pkt_base += 5u + pkt_used;
                    npk += 1;
                    pkt_used = 0;
                    pkt_first = ~0ull;
                    start_lane = f;
                    cum_before_start = cum_f;
                }
            }
        }
        if (pkt_used) {  // the last, partly filled packet
            if (lane == 0 && pkt_first != ~0ull) x.rec_wtag[pkt_first] = 0x80000000u | pkt_used;
            pkt_base += 5u + pkt_used;
            npk += 1;
        }
    }
    if (lane == 0) {
        x.conn_wlen[s] = pkt_base;
        x.conn_npk[s] = npk;
        if (ndropped) atomicAdd(x.n_dropped, ndropped);
    }
}

void launch_wire_layout(hipStream_t st, WorldDev w, WireDev x) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_wire_layout, dim3((w.S + 3) / 4), dim3(256), 0, st, w, x);
}

// Copy kernel.  A chunk of 64 consecutive records of one subscription segment is a CONTIGUOUS byte
// range of the connection's stream (~5-6 KB).  Each lane first describes its own message in LDS
// (start, header bytes incl. a packet tag if it opens a packet, payload pointer/length); then the wave
// writes the range cooperatively: lane l produces output dwords l, l+64, ... — message found by binary
// search of the 64 starts — so stores are fully coalesced and payload reads are 4-byte gathers from
// the L2-resident payload table.  Only the dwords that straddle a message / header boundary (and the
// unaligned ends of the range) are assembled byte by byte.
#define WIRE_HDR_MAX 32  // 5 tag + 1+3 + 1+5 + 2 + 1+3 + 1+3 = 25 bytes at most

struct WireChunk {
    uint32_t start[64];   // message start, relative to the chunk's first byte
    uint32_t hlen[64];    // header bytes (tag included)
    uint32_t plen[64];    // payload bytes
    const uint8_t *pay[64];
    __attribute__((aligned(16))) uint8_t hdr[64][WIRE_HDR_MAX];
};

__device__ __forceinline__ uint32_t put_varint(uint8_t *h, uint32_t n, uint32_t v) {
    while (v >= 0x80u) { h[n++] = (uint8_t)((v & 0x7Fu) | 0x80u); v >>= 7; }
    h[n++] = (uint8_t)v;
    return n;
}

// message index owning relative byte position r (messages are sorted, zero-length ones never own a byte)
__device__ __forceinline__ uint32_t owner_of(const WireChunk &c, uint32_t r) {
    uint32_t lo = 0, hi = 63;  // last j with start[j] <= r and (hlen+plen) > 0 reaching r
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (c.start[mid] <= r) lo = mid; else hi = mid - 1;
    }
    // zero-length (dropped / beyond the chunk) entries share the start of their successor: step back over them
    while (lo > 0 && c.hlen[lo] + c.plen[lo] == 0) lo--;
    return lo;
}

// which message owns relative byte r, and r's offset inside it.  When all 64 messages of the chunk have
// the same size T (the usual case: same payload type, same channel-id width, no packet tag in the chunk)
// that is a multiply-high by M = ceil(2^32 / T) — exact for r < 2^32 / T — instead of the search.
__device__ __forceinline__ void locate(const WireChunk &c, uint32_t r, uint32_t T, uint32_t M, uint32_t &j, uint32_t &k) {
    if (T) {
        j = __umulhi(r, M);
        k = r - j * T;
    } else {
        j = owner_of(c, r);
        k = r - c.start[j];
    }
}

__device__ __forceinline__ uint32_t byte_at(const WireChunk &c, uint32_t r, uint32_t T, uint32_t M) {
    uint32_t j, k;
    locate(c, r, T, M, j, k);
    return k < c.hlen[j] ? c.hdr[j][k] : c.pay[j][k - c.hlen[j]];
}


__global__ void __launch_bounds__(256) k_wire_copy(WorldDev w, WireDev x) {
    __shared__ uint32_t ticket;
    __shared__ WireChunk chunks[4];
    const uint32_t s = blockIdx.x;
    if (!w.sub_alive[s] || x.conn_woff[s + 1] == x.conn_woff[s]) return;  // (conn_wlen was scanned in place)
    const uint32_t lane = threadIdx.x & 63u;
    WireChunk &c = chunks[threadIdx.x >> 6];
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    const uint64_t rbase = w.rec_ub[s];
    uint8_t *stream = x.bytes + x.conn_woff[s];
    if (threadIdx.x == 0) ticket = 0;
    __syncthreads();
    for (;;) {
        uint32_t p = 0;
        if (lane == 0) p = atomicAdd(&ticket, 1u);
        p = __builtin_amdgcn_readfirstlane(p);
        if (p >= cnt) break;
        const uint32_t n = w.pair_nrec[pbase + p];
        const uint64_t seg = rbase + w.pair_rel[pbase + p];
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            // ---- describe this lane's message ----
            uint32_t begin = 0xFFFFFFFFu, hl = 0, pl = 0;
            const uint8_t *pay = nullptr;
            if (i < n) {
                const uint32_t woff = x.rec_woff[seg + i];
                if (woff != 0xFFFFFFFFu) {  // not dropped by the size check of Send
                    const WireMsg m = wire_msg(w, x, w.recs[seg + i], w.rec_pos[seg + i]);
                    const uint32_t tag = x.rec_wtag[seg + i];
                    uint8_t *h = c.hdr[lane];
                    begin = woff;
                    if (tag) {  // opens a packet: the 5-byte tag sits right before it (connection.go:683-687)
                        const uint32_t plen = tag & 0xFFFFu;
                        h[0] = 67; h[1] = 72; h[2] = (uint8_t)(plen >> 8); h[3] = (uint8_t)plen; h[4] = 0;
                        hl = 5;
                        begin -= 5;
                    }
                    h[hl++] = 0x0A; hl = put_varint(h, hl, m.mp_len);                     // Packet.messages
                    if (m.chan) { h[hl++] = 0x08; hl = put_varint(h, hl, m.chan); }       // MessagePack.channelId
                    h[hl++] = 0x20; h[hl++] = 0x08;                                       // MessagePack.msgType = CHANNEL_DATA_UPDATE
                    h[hl++] = 0x2A; hl = put_varint(h, hl, m.body_len);                   // MessagePack.msgBody
                    h[hl++] = 0x0A; hl = put_varint(h, hl, m.any_len);                    // ChannelDataUpdateMessage.data
                    pl = m.any_len;
                    pay = m.pay;
                }
            }
            // chunk range: first byte of the first live message .. end of the last
            uint32_t lo = begin, hi = begin == 0xFFFFFFFFu ? 0u : begin + hl + pl;
            for (int d = 32; d >= 1; d >>= 1) {
                lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
                hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
            }
            if (lo == 0xFFFFFFFFu) continue;  // nothing live in this chunk
            // dead entries take the start of the next live one (so that `start` stays sorted): suffix-min scan
            uint32_t st = begin;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_down(st, d);
                if (lane + d < 64) st = min(st, o);
            }
            if (st == 0xFFFFFFFFu) st = hi;
            c.start[lane] = st - lo;
            c.hlen[lane] = hl;
            c.plen[lane] = pl;
            c.pay[lane] = pay;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // uniform chunk?  all 64 lanes live, equal sizes
            const uint32_t mysz = hl + pl;
            const uint32_t T0 = __shfl(mysz, 0);
            const bool uniform = __ballot(begin == 0xFFFFFFFFu || mysz != T0) == 0 && T0 != 0;
            const uint32_t T = uniform ? T0 : 0u;
            const uint32_t M = uniform ? (uint32_t)((0x100000000ull + T0 - 1u) / T0) : 0u;
            // ---- write the range: dword t of the 4-byte-aligned cover goes to lane t % 64 ----
            uint8_t *dst0 = stream + lo;
            const uint32_t len = hi - lo;
            const uint32_t mis = (uint32_t)((uintptr_t)dst0 & 3u);  // bytes of the first dword that precede the range
            const uint32_t ndw = (mis + len + 3u) >> 2;
            for (uint32_t t = lane; t < ndw; t += 64) {
                const int32_t r0 = (int32_t)(t * 4u) - (int32_t)mis;  // relative position of the dword's first byte
                uint8_t *d = dst0 + r0;
                if (r0 >= 0 && (uint32_t)r0 + 4u <= len) {
                    uint32_t j, k;
                    locate(c, (uint32_t)r0, T, M, j, k);
                    const uint32_t h = c.hlen[j];
                    uint32_t v;
                    if (k >= h && k + 4u <= h + c.plen[j]) {
                        // payload interior: two aligned dwords of the (16-byte aligned) payload slot, funnel-shifted
                        const uint32_t o = k - h;
                        const uint32_t *src = (const uint32_t *)(const void *)(c.pay[j] + (o & ~3u));
                        const uint32_t a = src[0];
                        const uint32_t sh = o & 3u;
                        v = sh ? __builtin_amdgcn_alignbyte(src[1], a, sh) : a;
                    } else if (k + 4u <= h) {
                        // header interior: two aligned dwords of the lane's 32-byte header row in LDS, funnel-shifted
                        const uint32_t *hs = (const uint32_t *)(const void *)(c.hdr[j] + (k & ~3u));
                        const uint32_t sh = k & 3u;
                        v = sh ? __builtin_amdgcn_alignbyte(hs[1], hs[0], sh) : hs[0];
                    } else {
                        // straddles header/payload or two messages: byte by byte (two such dwords per message)
                        const uint32_t r = (uint32_t)r0;
                        v = byte_at(c, r, T, M) | (byte_at(c, r + 1u, T, M) << 8) | (byte_at(c, r + 2u, T, M) << 16) |
                            (byte_at(c, r + 3u, T, M) << 24);
                    }
                    *(uint32_t *)(void *)d = v;
                } else {
                    // an end of the range: only the bytes inside it (neighbouring chunks own the others)
                    for (int q = 0; q < 4; q++) {
                        const int32_t r = r0 + q;
                        if (r >= 0 && (uint32_t)r < len) d[q] = (uint8_t)byte_at(c, (uint32_t)r, T, M);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

void launch_wire_copy(hipStream_t st, WorldDev w, WireDev x) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_wire_copy, dim3(w.S), dim3(256), 0, st, w, x);
}

// payload upload: kind k, entry idx[i] <- lens[i] bytes at bytes + off[i]
__global__ void __launch_bounds__(256) k_wire_set_payloads(WireDev x, int full, int cell, uint32_t n, uint32_t limit,
                                                           const uint32_t *__restrict__ idx, const uint32_t *__restrict__ lens,
                                                           const uint64_t *__restrict__ off, const uint8_t *__restrict__ bytes) {
    const uint32_t u = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (u >= n) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = idx[u];
    if (i >= limit) return;
    const uint32_t len = lens[u];
    uint8_t *dst = (cell ? x.pay_cell[full] : x.pay_ent[full]) + (size_t)i * x.stride[full];
    const uint8_t *src = bytes + off[u];
    for (uint32_t k = lane; k < len; k += 64) dst[k] = src[k];
    if (lane == 0) (cell ? x.len_cell[full] : x.len_ent[full])[i] = len;
}

void launch_wire_set_payloads(hipStream_t st, WireDev x, int full, int cell, uint32_t n, uint32_t limit, const uint32_t *idx,
                              const uint32_t *lens, const uint64_t *off, const uint8_t *bytes) {
    if (!n) return;
    hipLaunchKernelGGL(k_wire_set_payloads, dim3((n + 3) / 4), dim3(256), 0, st, x, full, cell, n, limit, idx, lens, off, bytes);
}

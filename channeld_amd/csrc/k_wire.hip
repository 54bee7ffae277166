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
//   k_wire_copy    per chunk of 64 records (a contiguous byte range of the stream) the wave builds the bytes as an
//                  image in LDS — one lane per message: tag if it opens a packet, headers, payload — and streams
//                  the image out with 16-byte stores
// Bytes written per message = 5/packet + ~12 header + payload: this is the P*M term of SURVEY §8d.
#include "chd_kernels.h"

#define WIRE_MAX_PACKET 65535u
#define WIRE_DROP_SIZE 65530u  // MaxPacketSize - PacketHeaderSize (connection.go:72)

// the descriptor-driven build runs without a host round trip between sizing and writing: the writing kernels check the
// arenas themselves (the host re-runs them after growing an arena, chd_wire_build)
__device__ __forceinline__ bool wire_img_fits(const WorldDev &w, const WireDev &x) {
    return x.conn_woff[w.S] <= x.bytes_cap && x.rank_ndesc[w.S] <= x.cdesc_cap;
}

__device__ __forceinline__ uint32_t vlen(uint32_t v) { return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u; }


// ---------------------------------------------------------------------------
// TYPED merge of the buffered updates a window selected (data.go:225-269 + tpspb/data.go:227-252 + data.go:294):
// for the MOVEMENT subset of tpspb.EntityChannelData the accumulated update message is built field by field — proto.Merge
// on this schema: sub-messages merge recursively, a leaf with explicit presence is overwritten when the source has it — then
// filtered by the subscription's DataFieldMasks (fmutils.Filter) and written in field-number order, one-byte lengths (every
// nested length is below 128): the bytes Go's proto.Marshal produces.  Schema (unreal_common.proto:161-184, tps.proto:22-35):
//     EntityChannelData { 2: ActorState { 11: FRepMovement { 1 linearVelocity, 2 angularVelocity, 3 location, 4 rotation:
//     FVector { 1 x, 2 y, 3 z: float }; 5 bSimulatedPhysicSleep, 6 bRepPhysics: bool } } }
// An update that carries anything else (objRef, other states) is "not in the subset": the record then takes the generic
// path (the selected updates concatenated, which a protobuf parser reads as their merge).  oracle/merge.py restates this,
// pinned by python-protobuf on the reference's embedded descriptors (tests/golden/make_merge_golden.py).
// ---------------------------------------------------------------------------
#define CHD_SCHEMA_TPS_ENTITY_MOVEMENT 1u
#define MV_ACTOR 1u
#define MV_MOVE 2u
#define MV_VEC(f) (4u << (f))           // f = 0..3
#define MV_LEAF(f, a) (64u << ((f) * 3u + (a)))
#define MV_BOOL(k) (1u << (18u + (k)))  // present
#define MV_BVAL(k) (1u << (20u + (k)))  // value

struct Movement {
    uint32_t leaf[12];
    uint32_t bits;
};

__device__ __forceinline__ bool mv_varint(const uint8_t *__restrict__ b, uint32_t &i, uint32_t end, uint64_t &v) {
    v = 0;
    for (uint32_t s = 0; s < 64; s += 7) {
        if (i >= end) return false;
        const uint32_t c = b[i++];
        v |= (uint64_t)(c & 0x7Fu) << s;
        if (!(c & 0x80u)) return true;
    }
    return false;
}

// proto.Merge(m, parse(b[0..n))); false: not in the subset (m is then unusable)
__device__ bool mv_parse(const uint8_t *__restrict__ b, uint32_t n, Movement &m) {
    uint32_t i = 0;
    uint64_t v;
    while (i < n) {
        if (!mv_varint(b, i, n, v) || v != ((2u << 3) | 2u)) return false;           // actorState
        if (!mv_varint(b, i, n, v) || v > n - i) return false;
        const uint32_t e1 = i + (uint32_t)v;
        m.bits |= MV_ACTOR;
        while (i < e1) {
            if (!mv_varint(b, i, e1, v) || v != ((11u << 3) | 2u)) return false;      // replicatedMovement
            if (!mv_varint(b, i, e1, v) || v > e1 - i) return false;
            const uint32_t e2 = i + (uint32_t)v;
            m.bits |= MV_MOVE;
            while (i < e2) {
                if (!mv_varint(b, i, e2, v)) return false;
                const uint32_t fn = (uint32_t)(v >> 3), wt = (uint32_t)v & 7u;
                if (wt == 2u && fn >= 1u && fn <= 4u && v < 64u) {                  // an FVector
                    if (!mv_varint(b, i, e2, v) || v > e2 - i) return false;
                    const uint32_t e3 = i + (uint32_t)v, f = fn - 1u;
                    m.bits |= MV_VEC(f);
                    while (i < e3) {
                        if (!mv_varint(b, i, e3, v)) return false;
                        const uint32_t a = (uint32_t)(v >> 3) - 1u;
                        if (((uint32_t)v & 7u) != 5u || a > 2u || v >= 32u || e3 - i < 4u) return false;
                        const uint32_t val = (uint32_t)b[i] | ((uint32_t)b[i + 1] << 8) | ((uint32_t)b[i + 2] << 16) | ((uint32_t)b[i + 3] << 24);
                        i += 4;
                        // (constant indices: the leaves stay in registers)
#pragma unroll
                        for (uint32_t q = 0; q < 12; q++)
                            if (q == f * 3u + a) m.leaf[q] = val;
                        m.bits |= MV_LEAF(f, a);
                    }
                } else if (wt == 0u && (fn == 5u || fn == 6u) && v < 64u) {         // a bool
                    if (!mv_varint(b, i, e2, v)) return false;
                    const uint32_t k = fn - 5u;
                    m.bits = (m.bits | MV_BOOL(k)) & ~MV_BVAL(k);
                    if (v) m.bits |= MV_BVAL(k);
                } else {
                    return false;
                }
            }
        }
    }
    return true;
}

// fmutils.Filter with the subscription's DataFieldMasks in bit form (chd_sub_options.data_field_mask)
__device__ __forceinline__ void mv_filter(Movement &m, uint32_t fmask) {
    if (!fmask) return;
    if ((fmask & 64u) && !(fmask & 63u)) { m.bits = 0; return; }
#pragma unroll
    for (uint32_t f = 0; f < 4; f++)
        if (!((fmask >> f) & 1u)) m.bits &= ~(MV_VEC(f) | MV_LEAF(f, 0) | MV_LEAF(f, 1) | MV_LEAF(f, 2));
#pragma unroll
    for (uint32_t k = 0; k < 2; k++)
        if (!((fmask >> (4u + k)) & 1u)) m.bits &= ~(MV_BOOL(k) | MV_BVAL(k));
}

__device__ __forceinline__ uint32_t mv_move_len(const Movement &m) {
    uint32_t n = 0;
#pragma unroll
    for (uint32_t f = 0; f < 4; f++)
        if (m.bits & MV_VEC(f)) n += 2u + 5u * (uint32_t)__popc((m.bits >> (6u + 3u * f)) & 7u);
    return n + 2u * (uint32_t)__popc((m.bits >> 18) & 3u);
}
__device__ __forceinline__ uint32_t mv_size(const Movement &m) {
    if (!(m.bits & MV_ACTOR)) return 0u;
    return 2u + ((m.bits & MV_MOVE) ? 2u + mv_move_len(m) : 0u);
}

__device__ uint32_t mv_write(const Movement &m, uint8_t *d) {
    if (!(m.bits & MV_ACTOR)) return 0u;
    uint32_t n = 0;
    const uint32_t ml = mv_move_len(m);
    d[n++] = 0x12; d[n++] = (uint8_t)((m.bits & MV_MOVE) ? 2u + ml : 0u);
    if (m.bits & MV_MOVE) {
        d[n++] = 0x5A; d[n++] = (uint8_t)ml;
#pragma unroll
        for (uint32_t f = 0; f < 4; f++) {
            if (!(m.bits & MV_VEC(f))) continue;
            d[n++] = (uint8_t)(0x0Au + 8u * f);
            d[n++] = (uint8_t)(5u * (uint32_t)__popc((m.bits >> (6u + 3u * f)) & 7u));
#pragma unroll
            for (uint32_t a = 0; a < 3; a++) {
                if (!(m.bits & MV_LEAF(f, a))) continue;
                const uint32_t v = m.leaf[f * 3u + a];
                d[n++] = (uint8_t)(0x0Du + 8u * a);
                d[n++] = (uint8_t)v; d[n++] = (uint8_t)(v >> 8); d[n++] = (uint8_t)(v >> 16); d[n++] = (uint8_t)(v >> 24);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 2; k++)
            if (m.bits & MV_BOOL(k)) { d[n++] = (uint8_t)(0x28u + 8u * k); d[n++] = (m.bits & MV_BVAL(k)) ? 1 : 0; }
    }
    return n;
}

// the updates `mask` selects (bit j = tick cur - j, oldest first) of one entity channel, merged and filtered; false: one of
// them is not in the subset
__device__ bool mv_merge_selected(const WireDev &x, const uint8_t *__restrict__ ring_row, const uint32_t *__restrict__ rl, uint32_t mask,
                                  uint32_t fmask, Movement &m) {
    m.bits = 0;
#pragma unroll
    for (uint32_t q = 0; q < 12; q++) m.leaf[q] = 0;
    for (int jj = CHD_HIST_BITS - 1; jj >= 0; jj--) {
        if (!((mask >> jj) & 1u)) continue;
        const uint32_t sl = (x.cur_tick - (uint32_t)jj) & (CHD_HIST_BITS - 1u);
        if (!mv_parse(ring_row + (size_t)sl * x.stride[0], rl[sl], m)) return false;
    }
    mv_filter(m, fmask);
    return true;
}

struct WireMsg {
    uint32_t chan, any_len, body_len, mp_len, entry;  // entry = bytes inside the Packet (0: dropped by Send)
    const uint8_t *pay;                                // the one payload (Any bytes), or the channel's ring row in merge mode
    uint32_t mask;                                     // merge mode, update records: the ring slots to concatenate (bit j = tick cur - j)
    uint32_t value_len;                                // merge mode: bytes of Any.value (sum of the selected updates)
    uint32_t kind;                                     // merge mode: 1 + (0 entity, 1 cell) for an update record, 3 = entity update merged
                                                       // field by field (typed schema), else 0
};

// idc = CHD_POS_CELL | cell for a spatial channel's own message, else the entity's slot (k_wire_layout resolves the records'
// cell-table positions once and leaves idc in rec_pos for the copy kernels)
__device__ __forceinline__ WireMsg wire_msg(const WorldDev &w, const WireDev &x, chd_fanout_rec rec, uint32_t idc, uint32_t mask,
                                            uint32_t fmask = 0) {
    WireMsg m;
    const uint32_t full = rec.conn >> 31;
    m.chan = rec.channel;
    m.mask = 0; m.value_len = 0; m.kind = 0;
    const bool cell = (idc & CHD_POS_CELL) != 0;
    const uint32_t id = idc & ~CHD_POS_CELL;
    if (x.merge && !full) {
        // Any{type_url, value}: value = the buffered updates the window selected (data.go:225-269), oldest first
        const uint32_t *rl = (cell ? x.rlen_cell : x.rlen_ent) + (size_t)id * CHD_HIST_BITS;
        uint32_t total = 0;
        for (uint32_t mm = mask; mm; mm &= mm - 1u) total += rl[(x.cur_tick - (uint32_t)(__ffs((int)mm) - 1)) & (CHD_HIST_BITS - 1u)];
        const uint32_t ul = x.url_len[cell ? 1 : 0];
        m.mask = mask;
        m.value_len = total;
        m.kind = cell ? 2u : 1u;
        m.pay = (cell ? x.ring_cell : x.ring_ent) + (size_t)id * CHD_HIST_BITS * x.stride[0];
        if (!cell && x.schema == CHD_SCHEMA_TPS_ENTITY_MOVEMENT) {
            // the accumulated update message itself, as the reference marshals it (else: the selected updates concatenated)
            Movement mv;
            if (mv_merge_selected(x, m.pay, rl, mask, fmask, mv)) {
                total = mv_size(mv);
                m.value_len = total;
                m.kind = 3u;
            }
        }
        m.any_len = (ul ? 1u + vlen(ul) + ul : 0u) + (total ? 1u + vlen(total) + total : 0u);  // empty fields are not emitted
    } else if (cell) {
        m.any_len = x.len_cell[full][id];
        m.pay = x.pay_cell[full] + (size_t)id * x.stride[full];
    } else {
        m.any_len = x.len_ent[full][id];
        m.pay = x.pay_ent[full] + (size_t)id * x.stride[full];
    }
    m.body_len = 1u + vlen(m.any_len) + m.any_len;                       // ChannelDataUpdateMessage.data = 1
    m.mp_len = (m.chan ? 1u + vlen(m.chan) : 0u) + 2u                    // channelId = 1 (omitted if 0), msgType = 4 -> 0x20 0x08
               + 1u + vlen(m.body_len) + m.body_len;                     // msgBody = 5
    m.entry = m.mp_len >= WIRE_DROP_SIZE ? 0u : 1u + vlen(m.mp_len) + m.mp_len;  // Packet.messages = 1
    return m;
}

#define WIRE_FAST_PAY 80u  // k_wire_copy_fast keeps a message's payload in five 16-byte registers

// One wave per connection slot.
// Per record it leaves: rec_pos = idc (see wire_msg), rec_woff = offset of its Packet entry in the connection's stream
// (~0: dropped by Send), rec_wtag = any_len | packet length << 16 (packet length != 0 only for the record that opens a packet);
// per subscription segment seg_fast = 1 if every message of it carries at most WIRE_FAST_PAY payload bytes (no merging).
__global__ void __launch_bounds__(256) k_wire_layout(WorldDev w, WireDev x) {
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t pkt_base = 0;   // stream offset of the current packet's tag
    uint32_t pkt_used = 0;   // bytes of entries in the current packet
    uint64_t pkt_first = ~0ull;  // record index of the current packet's first entry
    uint32_t pkt_first_len = 0;  // ... and its any_len
    uint32_t npk = 0, ndropped = 0, any_slow = 0;
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        const size_t pbase = (size_t)s * w.capq;
        const uint64_t rbase = w.rec_ub[s];
        for (uint32_t p = 0; p < cnt; p++) {
            const uint32_t n = w.pair_nrec[pbase + p];
            const uint64_t seg = rbase + w.pair_rel[pbase + p];
            const uint32_t fmask = x.schema ? (w.pair_flags[pbase + p] >> PF_FIELD_MASK_SHIFT) & 0xFFu : 0u;  // DataFieldMasks, bit form
            bool fast = !x.merge && x.fast_ok;
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool valid = i < n;
                uint32_t entry = 0, alen = 0;
                if (valid) {
                    uint32_t pos = w.rec_pos[seg + i];
                    // (a record whose position word was never written would send the payload lookups anywhere: counted, and
                    // chd_wire_build fails, rather than a memory fault)
                    if ((pos & CHD_POS_CELL) ? (pos & ~CHD_POS_CELL) >= x.ncell : pos >= x.npos) {
                        atomicAdd(x.n_dropped + 1, 1u);
                        pos = CHD_POS_CELL;
                    }
                    uint32_t idc = (pos & CHD_POS_CELL) ? pos : w.ce_slot[pos];
                    if (!(idc & CHD_POS_CELL) && idc >= x.npay) {  // (same for a cell-table entry that names no entity slot)
                        atomicAdd(x.n_dropped + 2, 1u);
                        idc = CHD_POS_CELL;
                    }
                    const WireMsg m = wire_msg(w, x, w.recs[seg + i], idc, w.rec_mask ? w.rec_mask[seg + i] : 0u, fmask);
                    w.rec_pos[seg + i] = idc;
                    entry = m.entry;
                    alen = m.any_len;
                }
                if (__ballot(valid && alen > WIRE_FAST_PAY)) fast = false;
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
                        x.rec_wtag[seg + i] = alen;
                    }
                    // the first entry of the current packet, if it is in this range
                    const uint64_t firsts = __ballot(mine && lane < f && entry != 0);
                    if (pkt_first == ~0ull && firsts) {
                        const uint32_t fl = (uint32_t)__ffsll((unsigned long long)firsts) - 1u;
                        pkt_first = seg + i0 + fl;
                        pkt_first_len = (uint32_t)__shfl((int)alen, (int)fl);
                    }
                    const uint32_t cum_f = f < 64 ? __shfl(cum - entry, (int)f) : __shfl(cum, 63);  // prefix before lane f
                    pkt_used += cum_f - cum_before_start;
                    if (f == 64) break;
                    // close the packet (flush, connection.go:646-661): lane f's entry opens the next one
                    if (lane == 0 && pkt_first != ~0ull) x.rec_wtag[pkt_first] = (pkt_used << 16) | pkt_first_len;
                    pkt_base += 5u + pkt_used;
                    npk += 1;
                    pkt_used = 0;
                    pkt_first = ~0ull;
                    start_lane = f;
                    cum_before_start = cum_f;
                }
            }
            if (lane == 0) x.seg_fast[pbase + p] = (fast && n) ? 1 : 0;
            if (n && !fast) any_slow = 1;
        }
        if (pkt_used) {  // the last, partly filled packet
            if (lane == 0 && pkt_first != ~0ull) x.rec_wtag[pkt_first] = (pkt_used << 16) | pkt_first_len;
            pkt_base += 5u + pkt_used;
            npk += 1;
        }
    }
    if (lane == 0) {
        x.conn_wlen[s] = pkt_base;
        x.conn_npk[s] = npk;
        x.conn_slow[s] = (uint8_t)any_slow;  // k_wire_copy has segments of this connection to write
        if (any_slow) atomicAdd(x.n_dropped + 4, 1u);  // (how many connections: none -> the general copy kernel is not launched)
        if (ndropped) atomicAdd(x.n_dropped, ndropped);
    }
}

void launch_wire_layout(hipStream_t st, WorldDev w, WireDev x) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_wire_layout, dim3((w.S + 3) / 4), dim3(256), 0, st, w, x);
}

// Copy kernel.  A chunk of 64 consecutive records of one subscription segment is a CONTIGUOUS byte range of the
// connection's stream (~5 KB for position updates).  The wave builds that range as an IMAGE in LDS — every lane
// writes its own message there, header bytes then payload (16-byte loads from its channel's payload slot, unaligned
// LDS dword stores) — at the same misalignment modulo 16 as the destination, and then streams the image out with
// 16-byte LDS reads and 16-byte global stores, fully coalesced; only the two ends of the range are written byte by
// byte (the neighbouring chunks, other waves, own the adjacent bytes).  ~100 wave instructions per 64 messages.
// (The first version assigned output DWORDS to lanes and looked every one of them up — owner message, header or
// payload, two payload loads and a funnel shift per dword: ~2500 instructions per 64 messages, 0.5 TB/s.)
// A range longer than the image is cut into groups of consecutive messages; a single message longer than the image
// (a full state of tens of KB) is copied cooperatively straight from its payload slot.
#define WIRE_HDR_MAX 32  // 5 tag + 1+3 + 1+5 + 2 + 1+3 + 1+3 = 25 bytes at most
#define WIRE_IMG 8192u   // bytes of LDS image per wave

__device__ __forceinline__ uint32_t put_varint(uint8_t *h, uint32_t n, uint32_t v) {
    while (v >= 0x80u) { h[n++] = (uint8_t)((v & 0x7Fu) | 0x80u); v >>= 7; }
    h[n++] = (uint8_t)v;
    return n;
}

// bytes in front of the payload bytes proper: tag, the three nested headers and, in merge mode, the Any's own fields
__device__ __forceinline__ uint32_t wire_url_index(uint32_t kind) { return kind == 2u ? 1u : 0u; }  // (kinds 1 and 3: the entity data type)
__device__ __forceinline__ uint32_t wire_any_prefix(const WireDev &x, const WireMsg &m) {
    if (!m.kind) return 0u;
    const uint32_t ul = x.url_len[wire_url_index(m.kind)];
    return (ul ? 1u + vlen(ul) + ul : 0u) + (m.value_len ? 1u + vlen(m.value_len) : 0u);
}
__device__ __forceinline__ uint32_t wire_hdr_len(const WireDev &x, const WireMsg &m, uint32_t tag) {
    return (tag ? 5u : 0u) + 1u + vlen(m.mp_len) + (m.chan ? 1u + vlen(m.chan) : 0u) + 2u + 1u + vlen(m.body_len) + 1u + vlen(m.any_len) +
           wire_any_prefix(x, m);
}

// tag (if the message opens a packet) + the three nested length-delimited headers (+ Any.type_url and the key of Any.value
// in merge mode), byte by byte at h
__device__ __forceinline__ uint32_t put_header(uint8_t *h, const WireDev &x, const WireMsg &m, uint32_t tag) {
    uint32_t hl = 0;
    if (tag) {  // opens a packet: the 5-byte tag sits right before it (connection.go:683-687)
        const uint32_t plen = tag & 0xFFFFu;
        h[0] = 67; h[1] = 72; h[2] = (uint8_t)(plen >> 8); h[3] = (uint8_t)plen; h[4] = 0;
        hl = 5;
    }
    h[hl++] = 0x0A; hl = put_varint(h, hl, m.mp_len);                     // Packet.messages
    if (m.chan) { h[hl++] = 0x08; hl = put_varint(h, hl, m.chan); }       // MessagePack.channelId
    h[hl++] = 0x20; h[hl++] = 0x08;                                       // MessagePack.msgType = CHANNEL_DATA_UPDATE
    h[hl++] = 0x2A; hl = put_varint(h, hl, m.body_len);                   // MessagePack.msgBody
    h[hl++] = 0x0A; hl = put_varint(h, hl, m.any_len);                    // ChannelDataUpdateMessage.data
    if (m.kind) {
        const uint32_t ul = x.url_len[wire_url_index(m.kind)];
        if (ul) {                                                         // google.protobuf.Any.type_url = 1
            h[hl++] = 0x0A; hl = put_varint(h, hl, ul);
            const uint8_t *u = x.url[wire_url_index(m.kind)];
            for (uint32_t k = 0; k < ul; k++) h[hl++] = u[k];
        }
        if (m.value_len) { h[hl++] = 0x12; hl = put_varint(h, hl, m.value_len); }  // Any.value = 2
    }
    return hl;
}

// len bytes from a 16-byte aligned, padded global slot to d (LDS image, any alignment): 16-byte loads, unaligned dword stores
__device__ __forceinline__ void lane_put_payload(uint8_t *d, const uint8_t *__restrict__ pay, uint32_t len) {
    for (uint32_t q = 0; 16u * q < len; q++) {
        const uint4 v = *(const uint4 *)(const void *)(pay + 16u * q);
        const uint32_t rem = len - 16u * q;
        uint8_t *o = d + 16u * q;
        if (rem >= 16u) {
            __builtin_memcpy(o, &v.x, 4); __builtin_memcpy(o + 4, &v.y, 4);
            __builtin_memcpy(o + 8, &v.z, 4); __builtin_memcpy(o + 12, &v.w, 4);
        } else {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (4u * k + 4u <= rem) __builtin_memcpy(o + 4u * k, &vv[k], 4);
                else
#pragma unroll
                    for (uint32_t bb = 0; bb < 3; bb++)
                        if (4u * k + bb < rem) o[4u * k + bb] = (uint8_t)(vv[k] >> (8u * bb));
            }
        }
    }
}

__device__ __forceinline__ void wire_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// n bytes from src (any address space, any alignment) to global dst, by the whole wave: dwords where dst is aligned,
// bytes at the two ends
__device__ __forceinline__ void wave_copy_out(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t n, bool src_is_image) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t head = min((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u, n);
    if (lane < head) dst[lane] = src[lane];
    const uint32_t nvec = (n - head) >> 4;
    if (src_is_image) {  // same misalignment modulo 16 on both sides: 16-byte LDS reads, 16-byte stores
        for (uint32_t v = lane; v < nvec; v += 64)
            *(uint4 *)(void *)(dst + head + 16u * v) = *(const uint4 *)(const void *)(src + head + 16u * v);
    } else {             // global payload slot (16-byte aligned) to an arbitrary destination: funnel-shifted dwords
        const uint32_t ndw = nvec * 4u;
        const uint32_t sh = (uint32_t)((uintptr_t)(src + head) & 3u);
        const uint32_t *s32 = (const uint32_t *)(const void *)(src + head - sh);
        for (uint32_t t = lane; t < ndw; t += 64) {
            const uint32_t a = s32[t];
            *(uint32_t *)(void *)(dst + head + 4u * t) = sh ? __builtin_amdgcn_alignbyte(s32[t + 1], a, sh) : a;
        }
    }
    const uint32_t done = head + 16u * nvec;
    if (lane < n - done) dst[done + lane] = src[done + lane];
}

// One chunk of up to 64 messages, one per lane (live lanes: `begin` = byte offset of the message — of its tag, if it opens a
// packet — in `stream`, ascending with the lane; hl / pl = header and payload bytes; m, tag, fmask as wire_msg / put_header
// take them), written by the whole wave: groups of consecutive live messages whose byte range fits the LDS image are built
// there and streamed out; a single message longer than the image is copied cooperatively straight from its payload slot(s).
__device__ __forceinline__ void wire_put_messages(const WireDev &x, uint8_t *__restrict__ stream, uint8_t *img, bool live, uint32_t begin,
                                                  uint32_t hl, uint32_t pl, const WireMsg &m, uint32_t tag, uint32_t fmask) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t endb = begin + hl + pl;
    uint32_t a = 0;
    for (;;) {
        const uint64_t lm = __ballot(live && lane >= a);
        if (!lm) break;
        const uint32_t fa = (uint32_t)__ffsll((unsigned long long)lm) - 1u;
        const uint32_t lo = (uint32_t)__shfl((int)begin, (int)fa);
        const uint64_t nf = __ballot(live && lane >= fa && endb - lo > WIRE_IMG);
        const uint32_t b = nf ? (uint32_t)__ffsll((unsigned long long)nf) - 1u : 64u;
        if (b == fa) {
            // one message longer than the image: header through the image, payload straight from its slot(s)
            const uint32_t hl1 = (uint32_t)__shfl((int)hl, (int)fa);
            if (lane == fa) put_header(img, x, m, tag);
            wire_wave_sync();
            wave_copy_out(stream + lo, img, hl1, false);
            const uint32_t kind1 = (uint32_t)__shfl((int)m.kind, (int)fa), mask1 = (uint32_t)__shfl((int)m.mask, (int)fa);
            const uint64_t pay1 = ((uint64_t)(uint32_t)__shfl((int)((uintptr_t)m.pay >> 32), (int)fa) << 32) |
                                  (uint32_t)__shfl((int)(uint32_t)(uintptr_t)m.pay, (int)fa);
            if (!kind1) {
                wave_copy_out(stream + lo + hl1, (const uint8_t *)(uintptr_t)pay1, (uint32_t)__shfl((int)pl, (int)fa), false);
            } else {
                uint32_t at = lo + hl1;
                const uint32_t id1 = (uint32_t)__shfl((int)(((uintptr_t)m.pay - (uintptr_t)(kind1 == 2 ? x.ring_cell : x.ring_ent)) / ((size_t)CHD_HIST_BITS * x.stride[0])), (int)fa);
                const uint32_t *rl = (kind1 == 2 ? x.rlen_cell : x.rlen_ent) + (size_t)id1 * CHD_HIST_BITS;
                for (int jj = CHD_HIST_BITS - 1; jj >= 0; jj--) {
                    if (!((mask1 >> jj) & 1u)) continue;
                    const uint32_t sl = (x.cur_tick - (uint32_t)jj) & (CHD_HIST_BITS - 1u);
                    wave_copy_out(stream + at, (const uint8_t *)(uintptr_t)pay1 + (size_t)sl * x.stride[0], rl[sl], false);
                    at += rl[sl];
                }
            }
            wire_wave_sync();
            a = fa + 1;
            continue;
        }
        const bool mine = live && lane >= fa && lane < b;
        uint32_t hi = mine ? endb : 0u;
        for (int d = 32; d >= 1; d >>= 1) hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
        const uint32_t off0 = (uint32_t)((uintptr_t)(stream + lo) & 15u);
        if (mine) {
            uint8_t *d = img + off0 + (begin - lo);
            d += put_header(d, x, m, tag);
            if (!m.kind) {
                lane_put_payload(d, m.pay, pl);  // payload slots are 16-byte aligned and padded
            } else if (m.kind == 3u) {
                // the accumulated update message, merged field by field (the layout pass sized it the same way)
                const uint32_t id = (uint32_t)(((uintptr_t)m.pay - (uintptr_t)x.ring_ent) / ((size_t)CHD_HIST_BITS * x.stride[0]));
                Movement mv;
                if (mv_merge_selected(x, m.pay, x.rlen_ent + (size_t)id * CHD_HIST_BITS, m.mask, fmask, mv)) (void)mv_write(mv, d);
            } else {
                // the selected updates, oldest (highest bit) first
                const uint32_t *rl = (m.kind == 2 ? x.rlen_cell : x.rlen_ent) +
                                     ((uintptr_t)m.pay - (uintptr_t)(m.kind == 2 ? x.ring_cell : x.ring_ent)) / x.stride[0];
                for (int jj = CHD_HIST_BITS - 1; jj >= 0; jj--) {
                    if (!((m.mask >> jj) & 1u)) continue;
                    const uint32_t sl = (x.cur_tick - (uint32_t)jj) & (CHD_HIST_BITS - 1u);
                    const uint32_t ln = rl[sl];
                    lane_put_payload(d, m.pay + (size_t)sl * x.stride[0], ln);
                    d += ln;
                }
            }
        }
        wire_wave_sync();
        wave_copy_out(stream + lo, img + off0, hi - lo, true);
        wire_wave_sync();
        a = b;
    }
}

// the record-path segments (seg_fast == 0) of connection s, by one workgroup of four waves taking segments in turn
__device__ __forceinline__ void wire_copy_conn(const WorldDev &w, const WireDev &x, uint32_t s, uint8_t (*images)[WIRE_IMG + 48], uint32_t &ticket) {
    if (!w.sub_alive[s] || !x.conn_slow[s] || x.conn_woff[s + 1] == x.conn_woff[s]) return;  // (conn_wlen was scanned in place; uniform)
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *img = images[threadIdx.x >> 6];
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
        if (x.seg_fast[pbase + p]) continue;  // (k_wire_copy_fast writes those)
        const uint32_t n = w.pair_nrec[pbase + p];
        const uint64_t seg = rbase + w.pair_rel[pbase + p];
        const uint32_t fmask = x.schema ? (w.pair_flags[pbase + p] >> PF_FIELD_MASK_SHIFT) & 0xFFu : 0u;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            // ---- describe this lane's message ----
            bool live = false;
            uint32_t begin = 0, hl = 0, pl = 0, tag = 0;
            WireMsg m;
            m.chan = 0; m.any_len = 0; m.body_len = 0; m.mp_len = 0; m.entry = 0; m.pay = nullptr; m.mask = 0; m.value_len = 0; m.kind = 0;
            if (i < n) {
                const uint32_t woff = x.rec_woff[seg + i];
                if (woff != 0xFFFFFFFFu) {  // not dropped by the size check of Send
                    m = wire_msg(w, x, w.recs[seg + i], w.rec_pos[seg + i], w.rec_mask ? w.rec_mask[seg + i] : 0u, fmask);
                    tag = x.rec_wtag[seg + i] >> 16;  // packet length if the message opens a packet
                    if (tag) tag |= 0x80000000u;
                    hl = wire_hdr_len(x, m, tag);
                    pl = m.kind ? m.value_len : m.any_len;
                    begin = woff - (tag ? 5u : 0u);
                    live = true;
                }
            }
            wire_put_messages(x, stream, img, live, begin, hl, pl, m, tag, fmask);
        }
    }
}

__global__ void __launch_bounds__(256) k_wire_copy(WorldDev w, WireDev x) {
    __shared__ uint32_t ticket;
    __shared__ __attribute__((aligned(16))) uint8_t images[4][WIRE_IMG + 48];
    wire_copy_conn(w, x, blockIdx.x, images, ticket);
}

// the same over the LIST of connections that have such segments (k_wire_layout_img collects it): a fixed, small grid
__global__ void __launch_bounds__(256) k_wire_copy_list(WorldDev w, WireDev x) {
    __shared__ uint32_t ticket;
    __shared__ __attribute__((aligned(16))) uint8_t images[4][WIRE_IMG + 48];
    if (!wire_img_fits(w, x)) return;
    const uint32_t n = min(x.n_dropped[4], w.S);
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        wire_copy_conn(w, x, x.slow_list[k], images, ticket);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// The fast copy kernel: segments whose messages all carry <= WIRE_FAST_PAY payload bytes (k_wire_layout: seg_fast) — on a
// steady-state tick, all of them.  Same image technique as k_wire_copy, software-pipelined so that a wave never waits for
// its own record stores: on gfx950 the vm counter is in-order, and a wave that waits for ANY load also waits for every
// store it issued before it (~3.5 us while the chip streams).  Per chunk of 64 records the wave
//   - awaits the payload registers of THIS chunk and the per-record words of the NEXT one — loads issued one iteration
//     ago, BEFORE the previous chunk's stores: the wait is counted (vmcnt = the stores issued since), so those stores
//     stay in flight;
//   - builds the image in LDS (headers byte by byte, payload from registers);
//   - issues the payload loads of the next chunk and the per-record loads of the one after it;
//   - streams the image out.
// For the waits to be counted every global load and store of the loop is UNCONDITIONAL (straight-line code, a fixed
// number of instructions per iteration): lanes without a message load from a valid dummy address and store to a
// 16-byte trash slot of their wave.  Segment descriptors come from an LDS list built once per connection.
// ---------------------------------------------------------------------------
#define WIRE_FAST_IMG 7168u                     // 64 x (80 payload + 25 header + 5 tag) = 7040 bytes at most
#define WIRE_FAST_VECS (WIRE_FAST_IMG / 1024u)  // 16-byte vector stores per lane and chunk

typedef uint32_t wu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t wu32x2 __attribute__((ext_vector_type(2)));

struct WireFastMeta {
    wu32x2 rec;
    uint32_t idc, woff, wtag;
};

// Every global access of the loop is an ordinary load or store: the compiler tracks them and places the counted waits itself
// (and waits before it ever COPIES a register whose load is still in flight — a hand-placed s_waitcnt around inline-asm loads
// cannot: the register allocator is free to move an asm output to another register before the wait, and the hardware does not
// interlock a v_mov on an outstanding load; that version delivered garbage in ~0.04 % of the chunks).  What keeps its waits
// exact is the loop's shape: ONE entry with nothing in flight, a fixed number of unconditional loads and stores per step.
__device__ __forceinline__ void wire_fast_load_meta(const WorldDev &w, const WireDev &x, uint64_t base, uint32_t count, WireFastMeta &m) {
    // (lanes beyond the chunk re-read its last record: unconditional loads)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r = base + (lane < count ? lane : (count ? count - 1u : 0u));
    m.rec = *(const wu32x2 *)(const void *)(w.recs + r);
    m.idc = w.rec_pos[r];
    m.woff = x.rec_woff[r];
    m.wtag = x.rec_wtag[r];
}

// The payload of this lane's message: five 16-byte loads from its slot.  (Tried: the five pieces of a message fetched by
// ADJACENT lanes of one instruction — one cache-line request per message instead of five — with the pieces routed to the image
// by cross-lane reads: 34 more VGPRs, one wave per SIMD fewer, 8 % slower.  The kernel is bound by the image build — VALU and
// LDS instruction issue — not by L2 requests or HBM: with every store redirected to one trash line it takes the same time.)
__device__ __forceinline__ void wire_fast_load_pay(const uint8_t *pay, wu32x4 (&P)[WIRE_FAST_PAY / 16]) {
#pragma unroll
    for (uint32_t q = 0; q < WIRE_FAST_PAY / 16; q++) P[q] = *(const wu32x4 *)(const void *)(pay + 16u * q);
}

// (bases and strides come in as values: indexing the kernel argument's arrays with a per-lane `full` makes the compiler load
// them from memory inside the loop, and wait for that load with vmcnt(0))
__device__ __forceinline__ const uint8_t *wire_fast_pay(const WireFastMeta &m, const uint8_t *pe0, const uint8_t *pe1,
                                                        const uint8_t *pc0, const uint8_t *pc1, uint32_t st0, uint32_t st1) {
    const bool full = (m.rec.x >> 31) != 0;
    const uint32_t id = m.idc & ~CHD_POS_CELL;
    const uint8_t *base = (m.idc & CHD_POS_CELL) ? (full ? pc1 : pc0) : (full ? pe1 : pe0);
    return base + (size_t)id * (full ? st1 : st0);
}

#define WIRE_FAST_STORES (WIRE_FAST_VECS + 2u)  // VM stores per iteration: the vector stores + head bytes + tail bytes

__global__ void __launch_bounds__(256) k_wire_copy_fast(WorldDev w, WireDev x) {
    __shared__ uint32_t ticket, nseg;
    __shared__ uint32_t seg_n[512], seg_rel[512];
    __shared__ __attribute__((aligned(16))) uint8_t images[4][WIRE_FAST_IMG + 48];
    const uint32_t s = blockIdx.x;
    if (!w.sub_alive[s] || x.conn_woff[s + 1] == x.conn_woff[s]) return;  // (conn_wlen was scanned in place)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint8_t *img = images[wave];
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    const uint64_t rbase = w.rec_ub[s];
    uint8_t *stream = x.bytes + x.conn_woff[s];
    uint8_t *trash = x.trash + ((size_t)blockIdx.x * 4u + wave) * 16u;
    const uint64_t stream_len = x.conn_woff[s + 1] - x.conn_woff[s];
    // the connection's fast segments, compacted (order is irrelevant: every record knows its own offset)
    if (threadIdx.x == 0) { ticket = 0; nseg = 0; }
    __syncthreads();
    for (uint32_t p0 = 0; p0 < cnt; p0 += 256) {
        const uint32_t p = p0 + threadIdx.x;
        const bool f = p < cnt && x.seg_fast[pbase + p] == 1;
        if (f) {
            const uint32_t k = atomicAdd(&nseg, 1u);
            if (k < 512) { seg_n[k] = w.pair_nrec[pbase + p]; seg_rel[k] = w.pair_rel[pbase + p]; }
        }
    }
    __syncthreads();
    const uint32_t ns = min(nseg, 512u);  // (WireDev::fast_ok: capq <= 512)
    if (!ns) return;
    // ---- chunk cursor: (segment, offset) ----
    uint64_t cur_base = 0;
    uint32_t cur_n = 0, cur_i0 = 0;
    auto next_chunk = [&](uint64_t &base, uint32_t &count) {
        if (cur_i0 >= cur_n) {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&ticket, 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            if (k >= ns) { base = 0; count = 0; cur_n = 0; cur_i0 = 0; return; }
            cur_base = rbase + seg_rel[k];
            cur_n = seg_n[k];
            cur_i0 = 0;
        }
        base = cur_base + cur_i0;
        count = min(64u, cur_n - cur_i0);
        cur_i0 += 64;
    };
    const uint8_t *pe0 = x.pay_ent[0], *pe1 = x.pay_ent[1], *pc0 = x.pay_cell[0], *pc1 = x.pay_cell[1];
    const uint32_t st0 = x.stride[0], st1 = x.stride[1];
    uint64_t b0, b1;
    uint32_t c0, c1;
    next_chunk(b0, c0);
    if (!c0) return;
    // Two sets of per-record registers (X, Y) and of payload registers take turns, so that no value is copied while its
    // load is in flight (a copy would make the compiler wait right there).  The loop starts with a virtual EMPTY chunk
    // (nothing to build, every store to the trash slot) whose step issues the first real loads: one loop entry, nothing in
    // flight at it, and every step issues the same 9 loads and WIRE_FAST_STORES stores.
    WireFastMeta mA, mB;
    wu32x4 P0[WIRE_FAST_PAY / 16], P1[WIRE_FAST_PAY / 16];
    b1 = b0; c1 = c0;  // the first real chunk is the virtual one's "next"
    b0 = 0; c0 = 0;
    wire_fast_load_meta(w, x, b1, c1, mB);
    mA = mB;
#pragma unroll
    for (uint32_t q = 0; q < WIRE_FAST_PAY / 16; q++) { P0[q] = (wu32x4){0, 0, 0, 0}; P1[q] = (wu32x4){0, 0, 0, 0}; }
    asm volatile("" : "+v"(mA.rec), "+v"(mA.idc), "+v"(mA.woff), "+v"(mA.wtag), "+v"(mB.rec), "+v"(mB.idc), "+v"(mB.woff), "+v"(mB.wtag));  // (arrived)
    // one chunk: X = this chunk's per-record words, Y = the next chunk's, Pc = this chunk's payload registers, Pn = where the
    // next chunk's go; X is reloaded with the chunk after the next
    auto step = [&](WireFastMeta &X, WireFastMeta &Y, wu32x4 (&Pc)[WIRE_FAST_PAY / 16], wu32x4 (&Pn)[WIRE_FAST_PAY / 16]) {
        const uint32_t x_chan = X.rec.y, x_wtag = X.wtag, x_woff = X.woff;
        // ---- the loads of the following chunks go out NOW, under the image build below ----
        wire_fast_load_pay(lane < c1 && Y.woff != 0xFFFFFFFFu ? wire_fast_pay(Y, pe0, pe1, pc0, pc1, st0, st1) : pe0, Pn);
        uint64_t b2;
        uint32_t c2;
        next_chunk(b2, c2);
        wire_fast_load_meta(w, x, b2, c2, X);  // (x_chan, x_wtag, x_woff: X's arrived words, other registers from here on)
        // ---- this lane's message.  any_len <= 80, so every nested length is below 128 (one-byte varints) and the header is
        // [tag: 5] 0x0A mp_len | [0x08 varint(channel id): c = 2..6 bytes, absent for channel 0] | 0x20 0x08 0x2A body_len 0x0A any_len
        // = 8 + c bytes (put_header's layout; connection.go:57-83, data.go:293-318), composed in two 64-bit registers ----
        const bool live = lane < c0 && x_woff != 0xFFFFFFFFu;
        const uint32_t chan = x_chan, alen = x_wtag & 0xFFFFu, pk = x_wtag >> 16;
        const uint32_t nv = vlen(chan), c = chan ? 1u + nv : 0u;
        const uint32_t body = 2u + alen, mp = c + 4u + body;
        const uint32_t L = 8u + c, tagb = pk ? 5u : 0u;
        const uint32_t begin = live ? x_woff - tagb : 0xFFFFFFFFu;
        const uint32_t endb = live ? x_woff + L + alen : 0u;
        // (the layout pass numbers a chunk's messages in stream order: first live lane = lowest offset, last = highest end)
        const uint64_t lm = __ballot(live);
        const bool any = lm != 0;  // (uniform; a chunk of dropped messages only has nothing to write)
        const uint32_t lo = any ? (uint32_t)__builtin_amdgcn_readlane((int)begin, (int)__builtin_ctzll(lm | (1ull << 63))) : 0u;
        const uint32_t hi = any ? (uint32_t)__builtin_amdgcn_readlane((int)endb, 63 - (int)__builtin_clzll(lm | 1ull)) : 0u;
        uint32_t nbytes = hi - lo;
        if (hi < lo || nbytes > WIRE_FAST_IMG || (uint64_t)hi > stream_len) {  // (never, if the layout pass and this kernel agree)
            if (lane == 0) {
                const uint32_t k = atomicAdd(x.n_dropped + 3, 1u);
                if (k < 8) {  // (what the first few looked like: chd_wire_build's error message)
                    uint32_t *g = x.n_dropped + 8 + 8 * k;
                    g[0] = s; g[1] = lo; g[2] = hi; g[3] = (uint32_t)stream_len; g[4] = c0; g[5] = (uint32_t)lm; g[6] = (uint32_t)(lm >> 32); g[7] = x_woff;
                }
            }
            nbytes = 0;
        }
        const uint32_t off0 = any ? (uint32_t)((uintptr_t)(stream + lo) & 15u) : 0u;
        if (live) {
            // the payload first, in dwords (up to 3 bytes too many: the next message's header, written below by a LATER
            // instruction, covers them), then the headers, exactly
            uint8_t *pd = img + off0 + (begin - lo) + tagb + L;
#pragma unroll
            for (uint32_t q = 0; q < WIRE_FAST_PAY / 16; q++) {
                const uint32_t vv[4] = {Pc[q].x, Pc[q].y, Pc[q].z, Pc[q].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (16u * q + 4u * k < alen) __builtin_memcpy(pd + 16u * q + 4u * k, &vv[k], 4);
            }
        }
        if (live) {
            uint8_t *d = img + off0 + (begin - lo);
            if (pk) {  // opens a packet: the 5-byte tag sits right before it (connection.go:683-687)
                d[0] = 67; d[1] = 72; d[2] = (uint8_t)(pk >> 8); d[3] = (uint8_t)pk; d[4] = 0;
                d += 5;
            }
            uint64_t e = (uint64_t)((chan & 0x7Fu) | (((chan >> 7) & 0x7Fu) << 8) | (((chan >> 14) & 0x7Fu) << 16) | (((chan >> 21) & 0x7Fu) << 24)) |
                         ((uint64_t)(chan >> 28) << 32);
            e |= 0x8080808080ull & ((1ull << (8u * (nv - 1u))) - 1ull);  // continuation bits of all bytes but the last
            const uint64_t f3 = c ? (0x08ull | (e << 8)) : 0ull;
            const uint64_t f45 = 0x20ull | (0x08ull << 8) | (0x2Aull << 16) | ((uint64_t)body << 24) | (0x0Aull << 32) | ((uint64_t)alen << 40);
            const uint32_t sh = 16u + 8u * c;  // 16 .. 64
            const uint64_t lo64 = (0x0Aull | ((uint64_t)mp << 8)) | (f3 << 16) | (sh < 64u ? f45 << sh : 0ull);
            const uint64_t hi64 = sh > 16u ? f45 >> (64u - sh) : 0ull;
            const uint32_t w0 = (uint32_t)lo64, w1 = (uint32_t)(lo64 >> 32), w2 = (uint32_t)hi64, w3 = (uint32_t)(hi64 >> 32);
            __builtin_memcpy(d, &w0, 4);
            __builtin_memcpy(d + 4, &w1, 4);
            if (c >= 4u) __builtin_memcpy(d + 8, &w2, 4);
            const uint32_t tv = c >= 4u ? w3 : w2, tn = c & 3u;
            uint8_t *t = d + (c >= 4u ? 12 : 8);
            if (tn >= 1u) t[0] = (uint8_t)tv;
            if (tn >= 2u) t[1] = (uint8_t)(tv >> 8);
            if (tn >= 3u) t[2] = (uint8_t)(tv >> 16);
        }
        // ---- stream the image out: WIRE_FAST_VECS vector stores + head and tail bytes, all unconditional ----
        wire_wave_sync();
        {
            uint8_t *dst = stream + lo;
            const uint8_t *src = img + off0;
            const uint32_t head = min((16u - off0) & 15u, nbytes);
            const uint32_t nvec = (nbytes - head) >> 4;
            const uint32_t done = head + 16u * nvec;
            {
                const bool ok = lane < head;
                *(ok ? dst + lane : trash) = src[ok ? lane : 0u];
            }
#pragma unroll
            for (uint32_t j = 0; j < WIRE_FAST_VECS; j++) {
                const uint32_t v = lane + 64u * j;
                const bool ok = v < nvec;
                const wu32x4 val = *(const wu32x4 *)(const void *)(src + head + 16u * (ok ? v : 0u));
                *(wu32x4 *)(void *)(ok ? dst + head + 16u * v : trash) = val;
            }
            {
                const bool ok = lane < nbytes - done;
                *(ok ? dst + done + lane : trash) = src[ok ? done + lane : 0u];
            }
        }
        wire_wave_sync();
        b0 = b1; c0 = c1; b1 = b2; c1 = c2;
    };
    for (;;) {
        step(mA, mB, P0, P1);
        if (!c0) break;
        step(mB, mA, P1, P0);
        if (!c0) break;
    }
}

void launch_wire_copy(hipStream_t st, WorldDev w, WireDev x, uint32_t n_slow_conns) {
    if (!w.S) return;
    // segments of small messages (every steady-state update), then whatever is left (full states, merged updates) — if the
    // layout pass counted a connection with such a segment at all (a launch of S workgroups that all exit at once costs 0.7 ms
    // at 10 K connections)
    if (!x.merge && x.fast_ok) hipLaunchKernelGGL(k_wire_copy_fast, dim3(w.S), dim3(256), 0, st, w, x);
    if (n_slow_conns) hipLaunchKernelGGL(k_wire_copy, dim3(w.S), dim3(256), 0, st, w, x);
}

// ---------------------------------------------------------------------------
// Handover message assembly: what Notify builds for every handover (spatial.go:738-773,797-857).
// Two blobs per handover (entities with / without their full data, see include/chd_spatial.h); all nested lengths are
// computed bottom-up by ho_layout, the same routine sizes (k_handover_msg_sizes) and writes (k_handover_msg_write:
// one wave per blob, header bytes by lane 0, payloads copied by the wave).
// ---------------------------------------------------------------------------
#define WIRE_MSG_HANDOVER 12u  // MessageType_CHANNEL_DATA_HANDOVER (channeld.proto:133)

struct HoEnt {  // one entity of the handover: map entry {key = 1: netId, value = 2: SpatialEntityState{objRef = 1, entityData = 3}}
    uint32_t slot, net_id, ol, fl, state_len, entry_len;
};

__device__ __forceinline__ HoEnt ho_entity(const WorldDev &w, const WireDev &x, uint32_t slot, bool full) {
    HoEnt e;
    e.slot = slot;
    e.net_id = w.chan_id[slot];
    e.ol = x.len_objref[slot];
    e.fl = full ? x.len_ent[1][slot] : 0u;
    e.state_len = 1u + vlen(e.ol) + e.ol + (full ? 1u + vlen(e.fl) + e.fl : 0u);
    e.entry_len = (e.net_id ? 1u + vlen(e.net_id) : 0u) + 1u + vlen(e.state_len) + e.state_len;
    return e;
}

// members of handover h's entity list: the notifier alone, or the live members of its group
template <typename F>
__device__ __forceinline__ void ho_for_each_entity(const WorldDev &w, uint32_t notifier, F f) {
    const uint32_t gi = w.n_groups ? w.grp_of[notifier] : CHD_INVALID;
    if (gi == CHD_INVALID) { f(notifier); return; }
    for (uint32_t q = w.grp_off[gi]; q < w.grp_off[gi + 1]; q++) {
        const uint32_t m = w.grp_mem[q];
        if (w.eflags[m] & EF_ALIVE) f(m);
    }
}

struct HoLayout { uint32_t sd_len, any_len, hom_len, mp_len, ctx; };

// fullmask: bit q = entity q of the list carries its entityData (entities beyond 31 follow bit 31)
__device__ __forceinline__ bool ho_full(uint32_t fullmask, uint32_t q) { return (fullmask >> (q < 31u ? q : 31u)) & 1u; }

__device__ __forceinline__ HoLayout ho_layout(const DevGrid &g, const WorldDev &w, const WireDev &x, const chd_handover_rec &r, uint32_t fullmask) {
    HoLayout L;
    L.sd_len = 0;
    uint32_t q = 0;
    ho_for_each_entity(w, r.entity, [&](uint32_t m) {
        const HoEnt e = ho_entity(w, x, m, ho_full(fullmask, q++));
        L.sd_len += 1u + vlen(e.entry_len) + e.entry_len;  // SpatialChannelData.entities = 1
    });
    const uint32_t ul = x.url_len[2];
    L.any_len = (ul ? 1u + vlen(ul) + ul : 0u) + (L.sd_len ? 1u + vlen(L.sd_len) + L.sd_len : 0u);
    L.ctx = w.cell_sender[r.src - g.id_start];  // srcChannel.latestDataUpdateConnId (0 until the channel got an update)
    L.hom_len = 1u + vlen(r.src) + 1u + vlen(r.dst) + (L.ctx ? 1u + vlen(L.ctx) : 0u) + 1u + vlen(L.any_len) + L.any_len;
    L.mp_len = 1u + vlen(r.dst) + 2u + 1u + vlen(L.hom_len) + L.hom_len;  // channelId = dst, msgType = 12, msgBody
    return L;
}

__global__ void __launch_bounds__(256) k_handover_msg_sizes(DevGrid g, WorldDev w, WireDev x, uint32_t n_blobs, const uint32_t *__restrict__ var_h,
                                                            const uint32_t *__restrict__ var_mask, uint32_t *sizes) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= n_blobs) return;
    const uint32_t h = var_h ? var_h[b] : b >> 1, fullmask = var_h ? var_mask[b] : ((b & 1u) ? 0xFFFFFFFFu : 0u);
    sizes[b] = ho_layout(g, w, x, w.handovers[h], fullmask).mp_len;
}

__global__ void __launch_bounds__(64) k_handover_msg_write(DevGrid g, WorldDev w, WireDev x, uint32_t n_blobs, const uint32_t *__restrict__ var_h,
                                                           const uint32_t *__restrict__ var_mask, const uint32_t *__restrict__ off,
                                                           uint8_t *__restrict__ out, uint64_t cap) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blobs || off[b + 1] > cap) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t fullmask = var_h ? var_mask[b] : ((b & 1u) ? 0xFFFFFFFFu : 0u);
    const chd_handover_rec r = w.handovers[var_h ? var_h[b] : b >> 1];
    const HoLayout L = ho_layout(g, w, x, r, fullmask);
    uint8_t *o = out + off[b];
    uint32_t n = 0;
    if (lane == 0) {
        o[n++] = 0x08; n = put_varint(o, n, r.dst);                       // MessagePack.channelId = dstChannelId (spatial.go:771)
        o[n++] = 0x20; o[n++] = (uint8_t)WIRE_MSG_HANDOVER;               // MessagePack.msgType
        o[n++] = 0x2A; n = put_varint(o, n, L.hom_len);                   // MessagePack.msgBody
        o[n++] = 0x08; n = put_varint(o, n, r.src);                       // ChannelDataHandoverMessage.srcChannelId
        o[n++] = 0x10; n = put_varint(o, n, r.dst);                       // .dstChannelId
        if (L.ctx) { o[n++] = 0x18; n = put_varint(o, n, L.ctx); }        // .contextConnId
        o[n++] = 0x22; n = put_varint(o, n, L.any_len);                   // .data (Any)
        const uint32_t ul = x.url_len[2];
        if (ul) {
            o[n++] = 0x0A; n = put_varint(o, n, ul);                      // Any.type_url
            for (uint32_t k = 0; k < ul; k++) o[n++] = x.url[2][k];
        }
        if (L.sd_len) { o[n++] = 0x12; n = put_varint(o, n, L.sd_len); }  // Any.value = SpatialChannelData
    }
    n = (uint32_t)__shfl((int)n, 0);
    uint32_t q = 0;
    ho_for_each_entity(w, r.entity, [&](uint32_t m) {
        const bool full = ho_full(fullmask, q++);
        const HoEnt e = ho_entity(w, x, m, full);
        uint32_t k = n;
        if (lane == 0) {
            o[k++] = 0x0A; k = put_varint(o, k, e.entry_len);             // SpatialChannelData.entities (map entry)
            if (e.net_id) { o[k++] = 0x08; k = put_varint(o, k, e.net_id); }  // key
            o[k++] = 0x12; k = put_varint(o, k, e.state_len);             // value = SpatialEntityState
            o[k++] = 0x0A; k = put_varint(o, k, e.ol);                    // .objRef
        }
        k = (uint32_t)__shfl((int)k, 0);
        wave_copy_out(o + k, x.pay_objref + (size_t)m * x.stride[0], e.ol, false);
        k += e.ol;
        if (full) {
            uint32_t k2 = k;
            if (lane == 0) { o[k2++] = 0x1A; k2 = put_varint(o, k2, e.fl); }  // .entityData = Any of the full entity data
            k = (uint32_t)__shfl((int)k2, 0);
            wave_copy_out(o + k, x.pay_ent[1] + (size_t)m * x.stride[1], e.fl, false);
            k += e.fl;
        }
        n = k;
    });
}

void launch_handover_msg_sizes(hipStream_t st, DevGrid g, WorldDev w, WireDev x, uint32_t n_blobs, const uint32_t *var_h, const uint32_t *var_mask, uint32_t *sizes) {
    if (!n_blobs) return;
    hipLaunchKernelGGL(k_handover_msg_sizes, dim3((n_blobs + 255) / 256), dim3(256), 0, st, g, w, x, n_blobs, var_h, var_mask, sizes);
}
void launch_handover_msg_write(hipStream_t st, DevGrid g, WorldDev w, WireDev x, uint32_t n_blobs, const uint32_t *var_h, const uint32_t *var_mask,
                               const uint32_t *off, uint8_t *out, uint64_t cap) {
    if (!n_blobs) return;
    hipLaunchKernelGGL(k_handover_msg_write, dim3(n_blobs), dim3(64), 0, st, g, w, x, n_blobs, var_h, var_mask, off, out, cap);
}

// payload upload: kind k, entry idx[i] <- lens[i] bytes at bytes + off[i]
__global__ void __launch_bounds__(256) k_wire_set_payloads(WireDev x, int full, int cell, uint32_t n, uint32_t limit,
                                                           const uint32_t *__restrict__ idx, const uint32_t *__restrict__ lens,
                                                           const uint64_t *__restrict__ off, const uint8_t *__restrict__ bytes,
                                                           uint32_t ring_slot) {
    const uint32_t u = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (u >= n) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = idx[u];
    if (i >= limit) return;
    const uint32_t len = lens[u];
    uint8_t *dst;
    uint32_t *ldst;
    if (x.merge && full == 0) {  // this update's slot of the channel's ring
        const size_t e = (size_t)i * CHD_HIST_BITS + ring_slot;
        dst = (cell ? x.ring_cell : x.ring_ent) + e * x.stride[0];
        ldst = (cell ? x.rlen_cell : x.rlen_ent) + e;
    } else if (full == 2) {  // CHD_WIRE_ENTITY_OBJREF
        dst = x.pay_objref + (size_t)i * x.stride[0];
        ldst = x.len_objref + i;
    } else {
        dst = (cell ? x.pay_cell[full] : x.pay_ent[full]) + (size_t)i * x.stride[full];
        ldst = (cell ? x.len_cell[full] : x.len_ent[full]) + i;
    }
    const uint8_t *src = bytes + off[u];
    for (uint32_t k = lane; k < len; k += 64) dst[k] = src[k];
    if (lane == 0) *ldst = len;
}

void launch_wire_set_payloads(hipStream_t st, WireDev x, int full, int cell, uint32_t n, uint32_t limit, const uint32_t *idx,
                              const uint32_t *lens, const uint64_t *off, const uint8_t *bytes, uint32_t ring_slot) {
    if (!n) return;
    hipLaunchKernelGGL(k_wire_set_payloads, dim3((n + 3) / 4), dim3(256), 0, st, x, full, cell, n, limit, idx, lens, off, bytes, ring_slot);
}


// ---------------------------------------------------------------------------
// The wire streams from the fan-out DESCRIPTORS (k_fanout_plan_seg) instead of from the records.
//
// A simple subscription's messages are, per due window, "every entity channel of the cell, in column order" (+ the spatial
// channel's own message in front): the same byte sequence for every subscriber of the cell, apart from where the packet
// tags fall.  So the bytes are built ONCE per cell and tick — the cell's IMAGE: [own message][entity messages in column
// order], one image of the update payloads and one of the full states (first fan-outs) — and a connection's stream is a
// concatenation of image ranges with a 5-byte tag wherever the greedy packet rule (connection.go:626-714) cuts:
//
//   k_wire_img_sizes  one workgroup per cell and payload kind: entry size of every message (wire_msg), inclusive prefix
//                     -> img_end (message boundaries, the packet cuts fall on them), image length
//   scan              image lengths (16-byte aligned) -> image offsets in the arena
//   k_wire_img_fill   the image bytes (wire_put_messages: the same routine the record path writes streams with)
//   k_wire_layout_img one wave per connection walks its subscriptions in stream order — a handful of image ranges per
//                     subscription instead of ~300 records: packet cuts by a 64-ary search over the cell's message
//                     boundaries, one 16-byte copy descriptor {dst, src, len} per uncut range, the tags written directly.
//                     Runs twice: sizes only (stream lengths and descriptor counts -> scans), then emitting.
//                     Subscriptions without a usable descriptor (the deferred launch's, cells with an oversized message)
//                     are walked record by record as k_wire_layout does and left to k_wire_copy_list.
//   k_wire_conn_order the order the connections are copied in: by the cell of their first subscription
//   k_wire_copy_ends / k_wire_copy_img   the copy: unaligned 16-byte loads from the images (L2 resident: ~8 MB per tick at
//                     config B), destination-aligned 16-byte stores — the write stream is the only HBM traffic.
// No host round trip between sizing and writing: the writing kernels check the arenas on the device (wire_img_fits) and
// chd_wire_build re-runs them after growing one.
// Worlds with merged updates (per-record payloads) and ticks without descriptors keep the record path above.
// ---------------------------------------------------------------------------
#define SDW_NWIN 7u
#define SDW_FIRST 8u
#define SDW_NONE 16u
#define SDW_OWN_SHIFT 8u

// An image = (family f, column col, cell c): f = 0 the update payloads — col 0 every entity of the cell, col 1..9 the cell's
// WINDOW COLUMN col - 1 (partially updating ticks: the entities with an update inside that window shape, k_window_columns) —,
// f = 1 the full states (col 0 only).  Its k-th message is the entity at table position start + k (col 0) or the k-th entry of
// the window column (channel id in ce_chan's column, slot in wcol_slot's).
// In MERGE mode (CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS: a message carries the buffered updates its window selected) the
// update images of a cell are per WINDOW MASK instead: col 1..9 = every entity of the cell with the updates of
// wcol_mask(col - 1) merged — what a simple descriptor's window with that mask sends (k_fanout_plan_seg: the mask is the same
// for every entity of the cell) —, built only where a descriptor of the tick asks for it (img_need); no own message in front
// (a window with the spatial channel's own update is never simple in a masks world).
struct ImgSel { uint32_t f, col, c, start, n, ii, mask; bool by_column; size_t e0; };  // ii: index into img_len / img_own / img_bad / img_off; e0: into img_end

__device__ __forceinline__ uint32_t wcol_shape(uint32_t mask) {  // k with wcol_mask(k) == mask, else CHD_WCOLS
    uint32_t r = CHD_WCOLS;
#pragma unroll
    for (uint32_t k = 0; k < CHD_WCOLS; k++)
        if (mask == wcol_mask(k)) r = k;
    return r;
}

__device__ __forceinline__ ImgSel img_sel(const WorldDev &w, const WireDev &x, uint32_t c, uint32_t y, uint32_t ncol) {
    ImgSel s;
    s.f = y >= ncol ? 1u : 0u;
    s.col = s.f ? 0u : y;
    s.c = c;
    s.start = w.cell_start[c];
    s.n = w.cell_end[c] - s.start;
    s.mask = 0;
    s.by_column = false;
    s.ii = s.col * x.ncell + c;
    if (x.merge && !s.f) {
        s.mask = s.col ? wcol_mask(s.col - 1u) : 0u;
        if (!s.col || !x.img_need[s.ii]) s.n = 0;  // (column 0 has no meaning here)
    } else if (s.col) {
        // a column the cell's common history covers is never built (k_window_columns) nor referenced (k_fanout_plan_seg)
        s.n = (w.cell_hand[c] & wcol_mask(s.col - 1u)) ? 0u : w.cell_wcnt[(size_t)(s.col - 1u) * x.ncell + c];
        s.by_column = true;
    }
    s.e0 = (size_t)s.col * w.wcol_stride + s.start;
    return s;
}
__device__ __forceinline__ void img_entry(const WorldDev &w, const ImgSel &s, uint32_t k, uint32_t &slot, uint32_t &chan) {
    if (s.by_column) {
        const size_t at = (size_t)s.col * w.wcol_stride + s.start + k;
        chan = w.ce_chan[at];
        slot = w.wcol_slot[at];
    } else {
        chan = w.ce_chan_view[s.start + k];
        slot = w.ce_slot[s.start + k];
    }
}

// MERGE mode: which (mask, cell) images the tick's descriptors copy from.  One wave per connection, lanes over its descriptors.
__global__ void __launch_bounds__(256) k_wire_img_need(WorldDev w, WireDev x) {
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S || !w.sub_alive[s] || w.rec_ub[s + 1] > w.recs_cap) return;
    const uint32_t lane = threadIdx.x & 63u, ns = w.n_simple[s];
    const size_t pbase = (size_t)s * w.capq;
    for (uint32_t k = lane; k < ns; k += 64) {
        const uint32_t info = w.seg_desc[pbase + k].w, c = w.seg_desc2[pbase + k].x;
        const uint4 wm = w.seg_wm[pbase + k];
        const uint32_t m4[4] = {wm.x, wm.y, wm.z, wm.w};
        for (uint32_t j = 0; j < (info & SDW_NWIN); j++) {
            const uint32_t sh = wcol_shape(m4[j]);
            if (sh < CHD_WCOLS && sh + 1u < x.img_ncol) x.img_need[(size_t)(sh + 1u) * x.ncell + c] = 1u;
        }
    }
}

__global__ void __launch_bounds__(256) k_wire_img_sizes(DevGrid g, WorldDev w, WireDev x, uint32_t ncol) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t bad_any;
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const ImgSel sel = img_sel(w, x, c, blockIdx.y, ncol);
    const uint32_t f = sel.f, n = sel.n;
    if (tid == 0) bad_any = 0;
    // the spatial channel's own message
    chd_fanout_rec r;
    r.conn = f << 31;
    r.channel = c + g.id_start;
    const bool no_own = x.merge && !f;  // (merge mode: the update images carry no own message)
    WireMsg own = wire_msg(w, x, r, CHD_POS_CELL | c, 0u);
    if (no_own) own.entry = 0;
    uint32_t carry = own.entry;
    uint32_t bad = (own.entry == 0 && !no_own) ? 1u : 0u;  // (dropped by Send: the record path counts it)
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + tid;
        uint32_t entry = 0;
        if (i < n) {
            uint32_t slot, chan;
            img_entry(w, sel, i, slot, chan);
            if (slot >= x.npay) bad = 1;
            else {
                r.channel = chan;
                entry = wire_msg(w, x, r, slot, sel.mask).entry;
                if (!entry) bad = 1;
            }
        }
        uint32_t cum = entry;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(cum, d);
            if ((int)lane >= d) cum += o;
        }
        if (lane == 63) wsum[wv] = cum;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (k < wv) before += wsum[k];
            total += wsum[k];
        }
        if (i < n) x.img_end[f][sel.e0 + i] = carry + before + cum;
        carry += total;
        __syncthreads();
    }
    if (bad) bad_any = 1;
    __syncthreads();
    if (tid == 0) {
        const bool unused = sel.col != 0u && n == 0u;  // (no image: nothing references it)
        x.img_own[f][sel.ii] = own.entry;
        x.img_len[f][sel.ii] = unused ? 0u : carry;
        x.img_bad[f][sel.ii] = bad_any;
        x.img_off[f][sel.ii] = unused ? 0u : (carry + 15u) & ~15u;  // -> exclusive scan
    }
}

#define WIRE_FILL_SPLIT 4u
__global__ void __launch_bounds__(256) k_wire_img_fill(DevGrid g, WorldDev w, WireDev x, uint32_t ncol) {
    __shared__ __attribute__((aligned(16))) uint8_t images[4][WIRE_IMG + 48];
    // grid (cell, image * WIRE_FILL_SPLIT + part): the cell's chunks of 64 messages go round-robin over 4 * WIRE_FILL_SPLIT waves
    const uint32_t c = blockIdx.x, part = blockIdx.y % WIRE_FILL_SPLIT, lane = threadIdx.x & 63u;
    const ImgSel sel = img_sel(w, x, c, blockIdx.y / WIRE_FILL_SPLIT, ncol);
    const uint32_t f = sel.f, n = sel.n;
    const uint32_t wv = (threadIdx.x >> 6) + 4u * part;
    if (x.img_bad[f][sel.ii] || !x.img_len[f][sel.ii]) return;
    if (!x.img_need[f ? (size_t)x.img_ncol * x.ncell + c : sel.ii]) return;  // (no descriptor of the tick copies from it)
    const uint64_t off = x.img_off[f][sel.ii];
    if (off + x.img_len[f][sel.ii] > x.img_cap[f]) return;  // (never: the arena is sized for the worst case; the layout checks the same)
    uint8_t *stream = x.img[f] + off;
    uint8_t *img = images[threadIdx.x >> 6];
    chd_fanout_rec r;
    r.conn = f << 31;
    // chunk -1 = the own message (wave 0 starts with it), then the entity messages in chunks of 64, round-robin over the waves
    for (int32_t ch = (int32_t)wv - 1; ch * 64 < (int32_t)n; ch += 4 * (int32_t)WIRE_FILL_SPLIT) {
        bool live = false;
        uint32_t begin = 0, hl = 0, pl = 0;
        WireMsg m;
        m.chan = 0; m.any_len = 0; m.body_len = 0; m.mp_len = 0; m.entry = 0; m.pay = nullptr; m.mask = 0; m.value_len = 0; m.kind = 0;
        if (ch < 0) {
            if (lane == 0 && !(x.merge && !f)) {
                r.channel = c + g.id_start;
                m = wire_msg(w, x, r, CHD_POS_CELL | c, 0u);
                begin = 0;
                live = true;
            }
        } else {
            const uint32_t i = (uint32_t)ch * 64u + lane;
            if (i < n) {
                uint32_t slot;
                img_entry(w, sel, i, slot, r.channel);
                m = wire_msg(w, x, r, slot, sel.mask);
                begin = x.img_end[f][sel.e0 + i] - m.entry;
                live = true;
            }
        }
        if (live) {
            hl = wire_hdr_len(x, m, 0u);
            pl = m.kind ? m.value_len : m.any_len;
        }
        wire_put_messages(x, stream, img, live, begin, hl, pl, m, 0u, 0u);
    }
}

// number of boundaries E[0..n) (ascending) that are <= T: 64-ary search by the whole wave (uniform arguments and result)
__device__ __forceinline__ uint32_t wire_count_le(const uint32_t *__restrict__ E, uint32_t n, uint32_t T, uint32_t guess) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, hi = n;  // E[k] <= T for k < lo, E[k] > T for k >= hi
    {   // the messages of a cell are nearly equal in size: one probe of 64 consecutive boundaries around a linear guess
        // usually brackets the answer
        const uint32_t g0 = guess > 32u ? min(guess - 32u, n > 64u ? n - 64u : 0u) : 0u;
        const uint32_t idx = g0 + lane;
        const bool ok = idx < n && E[idx] <= T;
        const uint32_t mcount = (uint32_t)__popcll(__ballot(ok));
        const uint32_t span = min(64u, n - g0);
        if (mcount == 0u) hi = g0;                 // E[g0] > T
        else if (mcount == span) lo = g0 + span;   // all probed <= T
        else return g0 + mcount;
    }
    while (lo < hi) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + lane * step;
        const bool ok = idx < hi && E[idx] <= T;
        const uint32_t mcount = (uint32_t)__popcll(__ballot(ok));  // (monotone: the satisfied probes are a prefix)
        if (!mcount) { hi = lo; break; }
        const uint32_t nlo = lo + (mcount - 1u) * step + 1u;
        const uint32_t nhi = min(hi, lo + mcount * step);
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

// what a descriptor copies from: its cell, column and the images of both families (lane-private or uniform)
struct DescGeom { uint32_t info, c, cst, n, col, off0, len0, own0, off1, len1, own1; };

__device__ __forceinline__ bool desc_geom(const WorldDev &w, const WireDev &x, uint4 d, uint32_t c, DescGeom &G) {
    G.info = d.w; G.c = c; G.n = d.z;
    G.cst = w.cell_start[c];
    G.col = (w.wcol_on && w.wcol_stride) ? d.y / w.wcol_stride : 0u;
    G.off0 = G.len0 = G.own0 = G.off1 = G.len1 = G.own1 = 0;
    const uint32_t csz = w.cell_end[c] - G.cst;
    bool usable = d.y == G.cst + G.col * w.wcol_stride && G.col < x.img_ncol &&
                  d.z == (G.col ? w.cell_wcnt[(size_t)(G.col - 1u) * x.ncell + c] : csz);
    if (x.merge) {
        // merge mode: one image per window mask (checked per window: desc_windows_merge)
        if (G.col) usable = false;
    } else if (d.w & SDW_NWIN) {
        const uint32_t ii = G.col * x.ncell + c;
        if (!x.img_ok[0] || x.img_bad[0][ii]) usable = false;
        G.off0 = x.img_off[0][ii]; G.len0 = x.img_len[0][ii]; G.own0 = x.img_own[0][ii];
        if ((uint64_t)G.off0 + G.len0 > x.img_cap[0]) usable = false;
        if (G.col && (w.cell_hand[c] & wcol_mask(G.col - 1u))) usable = false;  // (a column that was not built)
    }
    if (d.w & SDW_FIRST) {
        if (G.col || !x.img_ok[1] || x.img_bad[1][c]) usable = false;
        G.off1 = x.img_off[1][c]; G.len1 = x.img_len[1][c]; G.own1 = x.img_own[1][c];
        if ((uint64_t)G.off1 + G.len1 > x.img_cap[1]) usable = false;
    }
    return usable;
}

// merge mode: window j of a descriptor copies the image of its mask (wm: the descriptor's seg_wm); false = not usable
__device__ __forceinline__ bool desc_window_merge(const WorldDev &w, const WireDev &x, uint32_t c, uint32_t mask, uint32_t &col, uint32_t &off, uint32_t &len) {
    const uint32_t sh = wcol_shape(mask);
    col = sh + 1u;
    off = len = 0;
    if (sh >= CHD_WCOLS || col >= x.img_ncol || !x.img_ok[0]) return false;
    const size_t ii = (size_t)col * x.ncell + c;
    if (!x.img_need[ii] || x.img_bad[0][ii]) return false;
    off = x.img_off[0][ii];
    len = x.img_len[0][ii];
    return (uint64_t)off + len <= x.img_cap[0];
}
__device__ __forceinline__ bool desc_windows_merge_ok(const WorldDev &w, const WireDev &x, uint32_t c, uint32_t info, uint4 wm) {
    const uint32_t m4[4] = {wm.x, wm.y, wm.z, wm.w};
    bool ok = !(info & SDW_NONE) || (info & SDW_NWIN) == 0u;
    for (uint32_t j = 0; j < (info & SDW_NWIN); j++) {
        uint32_t col, off, len;
        if ((info >> (SDW_OWN_SHIFT + j)) & 1u) ok = false;  // (never simple in a masks world)
        if (!desc_window_merge(w, x, c, m4[j], col, off, len)) ok = false;
    }
    return ok;
}

template <bool EMIT>
__global__ void __launch_bounds__(256) k_wire_layout_img(DevGrid g, WorldDev w, WireDev x) {
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t pkt_base = 0;  // stream offset of the current packet's tag
    uint32_t pkt_used = 0;  // bytes of entries in the current packet
    uint32_t npk = 0, ndropped = 0, any_slow = 0, ndesc = 0, stuck = 0;
    if (EMIT && !wire_img_fits(w, x)) return;
    const uint64_t woff = EMIT ? x.conn_woff[s] : 0ull;
    uint8_t *stream = EMIT ? x.bytes + woff : nullptr;
    uint4 *dout = EMIT ? x.cdesc + x.rank_ndesc[x.conn_rank[s]] : nullptr;  // (rank_ndesc was scanned in place)

    // A packet opened by a record of the record path: k_wire_copy writes contiguous byte ranges that span the messages of a
    // chunk — the tag's five bytes in front of that record lie inside such a range, so the record itself must carry the tag
    // (rec_wtag: packet length << 16, as k_wire_layout leaves it); the direct write below serves the packets opened by an
    // image range.
    uint64_t pkt_first = ~0ull;   // record that opened the current packet (~0: an image range did, or nothing yet)
    uint32_t pkt_first_len = 0;   // ... and its Any length (the low half of rec_wtag)
    bool pkt_started = false;     // the current packet holds an entry
    auto close_packet = [&]() {
        if (EMIT && lane == 0) {  // the 5-byte tag in front of the packet (connection.go:683-687)
            uint8_t *t = stream + pkt_base;
            t[0] = 67; t[1] = 72; t[2] = (uint8_t)(pkt_used >> 8); t[3] = (uint8_t)pkt_used; t[4] = 0;
            if (pkt_first != ~0ull) x.rec_wtag[pkt_first] = (pkt_used << 16) | pkt_first_len;
        }
        pkt_base += 5u + pkt_used;
        npk += 1;
        pkt_used = 0;
        pkt_first = ~0ull;
        pkt_started = false;
    };
    // The descriptors of a connection are contiguous, in stream order (consecutive ranges are adjacent in the stream: the
    // copy kernel's neighbouring waves fill whole 128-byte lines between them); the CONNECTIONS are ordered by the cell of
    // their first subscription (conn_rank), so the ~1000 connections being copied at any moment read a few cells' images.
    auto emit = [&](uint32_t f, uint64_t src, uint32_t len) {
        if (EMIT && lane == 0) {
            const uint64_t dst = woff + pkt_base + 5u + pkt_used;
            dout[ndesc] = make_uint4((uint32_t)dst, (uint32_t)(dst >> 32), (uint32_t)src | (f << 31), len);
        }
        ndesc += 1;
        pkt_used += len;
        pkt_started = true;
    };
    // bytes [a, b) of cell c's image f (off / own = image offset / own-message bytes; E = the entity messages' ends)
    auto piece = [&](uint32_t f, uint32_t off, uint32_t own, const uint32_t *__restrict__ E, uint32_t n, uint32_t a, uint32_t b, uint32_t ilen) {
        while (a < b) {
            const uint32_t room = WIRE_MAX_PACKET - pkt_used;
            if (b - a <= room) {
                emit(f, (uint64_t)off + a, b - a);
                break;
            }
            const uint32_t T = a + room;
            uint32_t e = a;
            if (a < own && own <= T) e = own;
            if (own <= T && n) {
                // (E[k] ~ own + (k + 1) * average entry)
                const uint32_t avg = max((ilen - own) / n, 1u);
                const uint32_t cnt = wire_count_le(E, n, T, T > own ? (T - own) / avg : 0u);
                if (cnt) {
                    const uint32_t ec = E[cnt - 1u];
                    if (ec > e && ec <= b) e = ec;
                }
            }
            if (e > a) emit(f, (uint64_t)off + a, e - a);
            if (!pkt_used) {  // (an entry is at most 65533 bytes: it fits an empty packet)
                stuck = 1;
                break;
            }
            close_packet();
            a = e;
        }
    };

    const bool served = w.sub_alive[s] && w.rec_ub[s + 1] <= w.recs_cap;
    if (served) {
        const uint32_t cnt = w.pair_cnt[s];
        const size_t pbase = (size_t)s * w.capq;
        const uint64_t rbase = w.rec_ub[s];
        for (uint32_t p0 = 0; p0 < cnt; p0 += 64) {
            const uint32_t p = p0 + lane;
            const uint32_t nrec = p < cnt ? w.pair_nrec[pbase + p] : 0u;
            const uint32_t rel = p < cnt ? w.pair_rel[pbase + p] : 0u;
            // the subscription's descriptors (k_fanout_plan_seg: pair_desc), the first one's geometry lane-parallel
            const uint32_t pd = (p < cnt && nrec) ? w.pair_desc[pbase + p] : 0xFFFFFFFFu;
            const bool simple = pd != 0xFFFFFFFFu;
            const uint32_t dfirst = pd & 0x0FFFFFFFu, ndp = simple ? pd >> 28 : 0u;
            DescGeom G;
            bool usable = false;
            if (simple) {
                usable = desc_geom(w, x, w.seg_desc[pbase + dfirst], w.seg_desc2[pbase + dfirst].x, G);
                if (w.seg_desc[pbase + dfirst].x != rel) usable = false;
                if (x.merge) {
                    if (!desc_windows_merge_ok(w, x, G.c, G.info, w.seg_wm[pbase + dfirst])) usable = false;
                    // DataFieldMasks make the typed merge the subscription's own: record path
                    if (x.schema && ((w.pair_flags[pbase + p] >> PF_FIELD_MASK_SHIFT) & 0xFFu)) usable = false;
                }
            }
            for (uint64_t act = __ballot(nrec != 0); act; act &= act - 1) {
                const int L = __ffsll((unsigned long long)act) - 1;
                const uint32_t ndL = (uint32_t)__builtin_amdgcn_readlane((int)ndp, L);
                const uint32_t dfL = (uint32_t)__builtin_amdgcn_readlane((int)dfirst, L);
                bool ok = __builtin_amdgcn_readlane((int)usable, L) != 0;
                // (a subscription split into one descriptor per window — partially updating ticks, windows that copy different
                // columns — is usable only if every part is: checked before anything is emitted)
                for (uint32_t t = 1; ok && t < ndL; t++) {
                    DescGeom Gt;
                    ok = desc_geom(w, x, w.seg_desc[pbase + dfL + t], w.seg_desc2[pbase + dfL + t].x, Gt);
                    if (ok && x.merge) ok = desc_windows_merge_ok(w, x, Gt.c, Gt.info, w.seg_wm[pbase + dfL + t]);
                }
                if (ok) {
                    for (uint32_t t = 0; t < ndL; t++) {
                        DescGeom H;
                        if (t == 0) {
                            H.info = (uint32_t)__builtin_amdgcn_readlane((int)G.info, L); H.cst = (uint32_t)__builtin_amdgcn_readlane((int)G.cst, L);
                            H.n = (uint32_t)__builtin_amdgcn_readlane((int)G.n, L); H.col = (uint32_t)__builtin_amdgcn_readlane((int)G.col, L);
                            H.off0 = (uint32_t)__builtin_amdgcn_readlane((int)G.off0, L); H.len0 = (uint32_t)__builtin_amdgcn_readlane((int)G.len0, L);
                            H.own0 = (uint32_t)__builtin_amdgcn_readlane((int)G.own0, L); H.off1 = (uint32_t)__builtin_amdgcn_readlane((int)G.off1, L);
                            H.len1 = (uint32_t)__builtin_amdgcn_readlane((int)G.len1, L); H.own1 = (uint32_t)__builtin_amdgcn_readlane((int)G.own1, L);
                            H.c = (uint32_t)__builtin_amdgcn_readlane((int)G.c, L);
                        } else {
                            (void)desc_geom(w, x, w.seg_desc[pbase + dfL + t], w.seg_desc2[pbase + dfL + t].x, H);
                        }
                        // (the sizing pass says which images the tick reads: k_wire_img_fill builds only those)
                        if (!EMIT && lane == 0) {
                            if (H.info & SDW_FIRST) x.img_need[(size_t)x.img_ncol * x.ncell + H.c] = 1u;
                            if (!x.merge && (H.info & SDW_NWIN)) x.img_need[(size_t)H.col * x.ncell + H.c] = 1u;
                        }
                        if (H.info & SDW_FIRST) piece(1u, H.off1, H.own1, x.img_end[1] + H.cst, H.n, 0u, H.len1, H.len1);
                        const uint32_t nw = H.info & SDW_NWIN;
                        if (x.merge) {
                            const uint4 wm = w.seg_wm[pbase + dfL + t];
                            const uint32_t m4[4] = {wm.x, wm.y, wm.z, wm.w};
                            for (uint32_t j = 0; j < nw; j++) {
                                uint32_t col, off, len;
                                (void)desc_window_merge(w, x, H.c, m4[j], col, off, len);
                                piece(0u, off, 0u, x.img_end[0] + (size_t)col * w.wcol_stride + H.cst, H.n, 0u, len, len);
                            }
                            continue;
                        }
                        const uint32_t *E0 = x.img_end[0] + (size_t)H.col * w.wcol_stride + H.cst;
                        for (uint32_t j = 0; j < nw; j++) {
                            const uint32_t a = ((H.info >> (SDW_OWN_SHIFT + j)) & 1u) ? 0u : H.own0;
                            const uint32_t b = (H.info & SDW_NONE) ? H.own0 : H.len0;
                            piece(0u, H.off0, H.own0, E0, H.n, a, b, H.len0);
                        }
                    }
                    if (EMIT && lane == 0) x.seg_fast[pbase + p0 + (uint32_t)L] = 2;  // (neither record-path copy kernel takes it)
                    continue;
                }
                // ---- record by record (k_wire_layout's walk; the tags are written here, so rec_wtag carries no packet length) ----
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)nrec, L);
                const uint64_t seg = rbase + (uint32_t)__builtin_amdgcn_readlane((int)rel, L);
                const uint32_t fmask = x.schema ? (w.pair_flags[pbase + p0 + (uint32_t)L] >> PF_FIELD_MASK_SHIFT) & 0xFFu : 0u;
                any_slow = 1;
                if (EMIT && lane == 0) x.seg_fast[pbase + p0 + (uint32_t)L] = 0;
                // whose message record i of the segment is.  The deferred launch left a position word per record; for a
                // subscription WITH descriptors (its images are not usable: a message Send drops, no arena) the record kernel
                // wrote none (WorldDev::seg_no_pos): the descriptors say it — per part the own message, then the column's entries
                auto derived = [&](uint32_t i) -> uint32_t {
                    uint32_t acc = 0, out = 0xFFFFFFFFu;
                    for (uint32_t t = 0; t < ndL; t++) {
                        const uint4 d = w.seg_desc[pbase + dfL + t];
                        const uint32_t c = w.seg_desc2[pbase + dfL + t].x, cst = w.cell_start[c], nn = d.z;
                        const uint32_t col = (w.wcol_on && w.wcol_stride) ? d.y / w.wcol_stride : 0u;
                        auto ent = [&](uint32_t k) { return col ? w.wcol_slot[(size_t)col * w.wcol_stride + cst + k] : w.ce_slot[cst + k]; };
                        if (d.w & SDW_FIRST) {
                            if (i == acc) out = CHD_POS_CELL | c;
                            else if (i > acc && i <= acc + nn) out = ent(i - acc - 1u);
                            acc += nn + 1u;
                        }
                        for (uint32_t j = 0; j < (d.w & SDW_NWIN); j++) {
                            if ((d.w >> (SDW_OWN_SHIFT + j)) & 1u) {
                                if (i == acc) out = CHD_POS_CELL | c;
                                acc += 1u;
                            }
                            if (!(d.w & SDW_NONE)) {
                                if (i >= acc && i < acc + nn) out = ent(i - acc);
                                acc += nn;
                            }
                        }
                    }
                    return out;
                };
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    const bool valid = i < n;
                    uint32_t entry = 0, alen = 0;
                    if (valid) {
                        uint32_t pos = ndL ? 0u : w.rec_pos[seg + i];
                        if ((pos & CHD_POS_CELL) ? (pos & ~CHD_POS_CELL) >= x.ncell : pos >= x.npos) {
                            if (EMIT) atomicAdd(x.n_dropped + 1, 1u);
                            pos = CHD_POS_CELL;
                        }
                        uint32_t idc = ndL ? derived(i) : (pos & CHD_POS_CELL) ? pos : w.ce_slot[pos];
                        if (ndL && (idc & CHD_POS_CELL) && (idc & ~CHD_POS_CELL) >= x.ncell) {  // (no part of the descriptors claims the record)
                            if (EMIT) atomicAdd(x.n_dropped + 1, 1u);
                            idc = CHD_POS_CELL;
                        }
                        if (!(idc & CHD_POS_CELL) && idc >= x.npay) {
                            if (EMIT) atomicAdd(x.n_dropped + 2, 1u);
                            idc = CHD_POS_CELL;
                        }
                        const WireMsg m = wire_msg(w, x, w.recs[seg + i], idc, w.rec_mask ? w.rec_mask[seg + i] : 0u, fmask);
                        if (EMIT) w.rec_pos[seg + i] = idc;  // (k_wire_copy reads the resolved word; the sizing pass must leave it alone)
                        entry = m.entry;
                        alen = m.any_len;
                    }
                    if (EMIT && valid && entry == 0) x.rec_woff[seg + i] = 0xFFFFFFFFu;  // dropped
                    ndropped += (uint32_t)__popcll(__ballot(valid && entry == 0));
                    uint32_t cum = entry;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o = __shfl_up(cum, d);
                        if ((int)lane >= d) cum += o;
                    }
                    uint32_t start_lane = 0, cum_before_start = 0;
                    for (;;) {
                        const uint32_t relb = pkt_used + (cum - cum_before_start);
                        const bool mine = valid && lane >= start_lane;
                        const uint64_t over = __ballot(mine && entry != 0 && relb > WIRE_MAX_PACKET);
                        const uint32_t fo = over ? (uint32_t)__ffsll((unsigned long long)over) - 1u : 64u;
                        if (EMIT && mine && lane < fo && entry != 0) {
                            x.rec_woff[seg + i] = (uint32_t)(pkt_base + 5u + relb - entry);
                            x.rec_wtag[seg + i] = alen;
                        }
                        const uint64_t firsts = __ballot(mine && lane < fo && entry != 0);
                        if (firsts) {
                            if (!pkt_started) {
                                const uint32_t fl = (uint32_t)__ffsll((unsigned long long)firsts) - 1u;
                                pkt_first = seg + i0 + fl;
                                pkt_first_len = (uint32_t)__shfl((int)alen, (int)fl);
                            }
                            pkt_started = true;
                        }
                        const uint32_t cum_f = fo < 64 ? __shfl(cum - entry, (int)fo) : __shfl(cum, 63);
                        pkt_used += cum_f - cum_before_start;
                        if (fo == 64) break;
                        close_packet();
                        start_lane = fo;
                        cum_before_start = cum_f;
                    }
                }
            }
        }
        if (pkt_used) close_packet();
    }
    if (lane == 0) {
        if (!EMIT) {
            x.conn_wlen[s] = pkt_base;
            x.conn_ndesc[s] = ndesc;
            // place among the connections: by the cell of the first subscription, arrival order inside a cell
            const uint32_t key = (served && w.pair_cnt[s]) ? w.pair_cell[(size_t)s * w.capq] : 0u;
            x.conn_key[s] = key;
            x.conn_rank[s] = atomicAdd(&x.cell_dcnt[(size_t)key * x.dpad], 1u);  // (-> k_wire_conn_order adds the cell's base)
            x.conn_npk[s] = npk;
            x.conn_slow[s] = (uint8_t)any_slow;
            if (any_slow) x.slow_list[atomicAdd(x.n_dropped + 4, 1u)] = s;
            if (ndropped) atomicAdd(x.n_dropped, ndropped);
            if (stuck) atomicAdd(x.n_dropped + 6, 1u);
        }
    }
}

typedef uint32_t wu32x4u __attribute__((ext_vector_type(4), aligned(1)));

// The copy: persistent workgroups take tickets of WIRE_CP_BATCH consecutive descriptors (half a connection's stream: ranges
// that are adjacent in the stream are written close in time, so the 128-byte lines they share leave L2 whole).  It reads
// as many bytes as it writes, and the reads should stay in L2: what was measured at config B (5.03 GB per build, whole
// build incl. layout): connections in slot order 1.87 ms — the 2048 waves read all ~8 MB of images at once; every read
// confined to a 1 MB window (wrong bytes, a bound) 1.34 ms; cells split over the XCDs' L2s by HW_REG_XCC_ID 1.74 ms;
// descriptors sorted by CELL (one or two images live at a time, but the stream written in 18 KB pieces scattered in time:
// partial lines) 1.63 ms; connections ordered by the cell of their first subscription (conn_rank) 1.51 ms — what is built.
// k_wire_copy_ends writes the up to 15 bytes in front of and behind the 16-byte aligned body of every range;
// k_wire_copy_img moves the bodies.
#define WIRE_CP_BATCH 16u

struct CpRange { const uint8_t *sp; uint8_t *dp; uint32_t nvec; };

__device__ __forceinline__ CpRange cp_range(const WireDev &x, uint4 d) {
    uint8_t *dst = x.bytes + (((uint64_t)d.y << 32) | d.x);
    const uint8_t *src = x.img[d.z >> 31] + (d.z & 0x7FFFFFFFu);
    const uint32_t head = min((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u, d.w);
    CpRange r;
    r.sp = src + head;
    r.dp = dst + head;
    r.nvec = (d.w - head) >> 4;
    return r;
}

// 32 threads per range: thread j < 16 the j-th byte in front of the body, thread 16 + j the j-th byte behind it (one
// thread per range walking up to 30 bytes took 54 us: its byte loads and stores serialise)
__global__ void __launch_bounds__(256) k_wire_copy_ends(WorldDev w, WireDev x) {
    if (!wire_img_fits(w, x)) return;
    const uint32_t ndesc = x.rank_ndesc[w.S];
    const uint32_t j = threadIdx.x & 31u;
    for (uint32_t i = blockIdx.x * 8u + (threadIdx.x >> 5); i < ndesc; i += gridDim.x * 8u) {
        const uint4 d = x.cdesc[i];
        uint8_t *dst = x.bytes + (((uint64_t)d.y << 32) | d.x);
        const uint8_t *src = x.img[d.z >> 31] + (d.z & 0x7FFFFFFFu);
        const uint32_t len = d.w;
        const uint32_t head = min((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u, len);
        const uint32_t done = head + (((len - head) >> 4) << 4);
        if (j < 16u) {
            if (j < head) dst[j] = src[j];
        } else if (done + (j - 16u) < len) {
            dst[done + (j - 16u)] = src[done + (j - 16u)];
        }
    }
}

// k_wire_copy_img: workgroups of two LOADER and two STORER waves around double-buffered LDS tiles.  gfx950's vm counter is
// in-order — a wave that waits for a load also waits for every store it issued before (~3.5 us while the chip streams) — so
// the waves that store never load: loader p reads piece after piece (CPW_K 16-byte vectors per lane, unaligned: the body is
// aligned to the DESTINATION) from the images into its tile, storer p drains the tile of the phase before into the stream
// with aligned 16-byte stores that nothing ever waits for.  One LDS-only barrier per phase (lds_barrier: __syncthreads()
// would wait for the stores).  All four waves walk the same sequence of pieces — the bodies of the ticket's ranges, cut
// into pieces of at most 64 * CPW_K vectors — from the descriptors each holds in its registers.
// (A single-wave version with counted s_waitcnt around inline-asm loads was dropped: the register allocator copied
// in-flight registers across the branches that chose the count.)
#ifndef CPW_K
#define CPW_K 4u
#endif
#define CPW_PIECE (64u * CPW_K)

__device__ __forceinline__ void cpw_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct CpStream {  // the ticket's pieces in order: uniform per wave
    uint4 dl;
    uint64_t todo;
    CpRange r;
    uint32_t v;     // next vector of r
    bool open;
};

__device__ __forceinline__ bool cpw_next(const WireDev &x, CpStream &st, const uint8_t *&sp, uint8_t *&dp, uint32_t &nv) {
    if (!st.open || st.v >= st.r.nvec) {
        if (!st.todo) { st.open = false; return false; }
        const int L = __ffsll((unsigned long long)st.todo) - 1;
        st.todo &= st.todo - 1;
        uint4 d;
        d.x = (uint32_t)__builtin_amdgcn_readlane((int)st.dl.x, L); d.y = (uint32_t)__builtin_amdgcn_readlane((int)st.dl.y, L);
        d.z = (uint32_t)__builtin_amdgcn_readlane((int)st.dl.z, L); d.w = (uint32_t)__builtin_amdgcn_readlane((int)st.dl.w, L);
        st.r = cp_range(x, d);
        st.v = 0;
        st.open = true;
    }
    sp = st.r.sp + 16u * (size_t)st.v;
    dp = st.r.dp + 16u * (size_t)st.v;
    nv = min(st.r.nvec - st.v, CPW_PIECE);
    st.v += CPW_PIECE;
    return true;
}

template <bool NT>
__global__ void __launch_bounds__(256) k_wire_copy_img(WorldDev w, WireDev x) {
    __shared__ wu32x4 tile[2][2][CPW_PIECE];
    __shared__ uint32_t ticket_s;
    if (!wire_img_fits(w, x)) return;
    const uint32_t ndesc = x.rank_ndesc[w.S];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), pair = wv & 1u;  // (scalar: the roles are separate code paths)
    const bool storer = wv >= 2u;
    for (;;) {
        if (threadIdx.x == 0) ticket_s = atomicAdd(x.cp_ticket, 1u);
        cpw_lds_barrier();
        const uint32_t i0 = ticket_s * WIRE_CP_BATCH;
        cpw_lds_barrier();  // (everyone has read the ticket before thread 0 takes the next one)
        if (i0 >= ndesc) break;
        CpStream st;
        st.dl = make_uint4(0u, 0u, 0u, 0u);
        const bool have = lane < WIRE_CP_BATCH && i0 + lane < ndesc;
        if (have) st.dl = x.cdesc[i0 + lane];
        const uint32_t my_nvec = have ? cp_range(x, st.dl).nvec : 0u;
        st.todo = __ballot(my_nvec != 0u);
        st.open = false;
        st.v = 0;
        uint32_t np = (my_nvec + CPW_PIECE - 1u) / CPW_PIECE;  // pieces of the ticket
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) np += (uint32_t)__shfl_xor((int)np, d);
        const uint32_t fills = (np + 1u) / 2u;
        // phase ph < fills: the loaders fill tile[ph & 1] with pieces 2 ph (pair 0) and 2 ph + 1 (pair 1);
        // phase ph >= 1:   the storers drain tile[(ph - 1) & 1].  One barrier per phase, fills + 1 phases.
        for (uint32_t ph = 0; ph <= fills; ph++) {
            const bool work = storer ? ph >= 1u : ph < fills;
            if (work) {
                const uint8_t *sp = nullptr, *sp2 = nullptr;
                uint8_t *dp = nullptr, *dp2 = nullptr;
                uint32_t nv = 0, nv2 = 0;
                const bool a = cpw_next(x, st, sp, dp, nv);
                const bool b2 = a && cpw_next(x, st, sp2, dp2, nv2);
                if (pair) { sp = sp2; dp = dp2; nv = b2 ? nv2 : 0u; }
                else if (!a) nv = 0u;
                wu32x4 q[CPW_K];
                if (nv) {  // (uniform)
                    if (!storer) {
                        // unconditional loads: a lane beyond the piece repeats its last vector (in bounds, into its own tile slot)
#pragma unroll
                        for (uint32_t k = 0; k < CPW_K; k++) q[k] = *(const wu32x4u *)(const void *)(sp + 16u * (size_t)min(lane + 64u * k, nv - 1u));
#pragma unroll
                        for (uint32_t k = 0; k < CPW_K; k++) tile[ph & 1u][pair][lane + 64u * k] = q[k];
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < CPW_K; k++) q[k] = tile[(ph - 1u) & 1u][pair][lane + 64u * k];
#pragma unroll
                        for (uint32_t k = 0; k < CPW_K; k++)
                            if (lane + 64u * k < nv) {
                                if (NT) __builtin_nontemporal_store(q[k], (wu32x4 *)(void *)(dp + 16u * (size_t)(lane + 64u * k)));
                                else *(wu32x4 *)(void *)(dp + 16u * (size_t)(lane + 64u * k)) = q[k];
                            }
                    }
                }
            }
            cpw_lds_barrier();
        }
    }
}

// rank of every connection in the copy order (cell_dcnt: connections per first cell, scanned) and its descriptor count there
__global__ void __launch_bounds__(256) k_wire_conn_order(WireDev x, uint32_t S) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= S) return;
    const uint32_t r = x.cell_dcnt[(size_t)x.conn_key[s] * x.dpad] + x.conn_rank[s];
    x.conn_rank[s] = r;
    x.rank_ndesc[r] = x.conn_ndesc[s];
}

void launch_wire_conn_order(hipStream_t st, WorldDev w, WireDev x) {
    if (!w.S) return;
    launch_scan_u32_inplace(st, x.cell_dcnt, x.ncell * x.dpad);
    hipLaunchKernelGGL(k_wire_conn_order, dim3((w.S + 255u) / 256u), dim3(256), 0, st, x, w.S);
    launch_scan_u32_inplace(st, x.rank_ndesc, w.S);
}

static uint32_t wire_image_columns(const WorldDev &w, const WireDev &x) {
    // update images per cell: the full column (+ the window columns of a partially updating tick; merge mode: one per window mask)
    return (w.wcol_on || x.merge) ? x.img_ncol : 1u;
}

// the message boundaries and lengths of every image (the layout needs them) ...
void launch_wire_images(hipStream_t st, DevGrid g, WorldDev w, WireDev x) {
    if (!g.ncell) return;
    const uint32_t ncol = wire_image_columns(w, x);
    (void)hipMemsetAsync(x.img_need, 0, sizeof(uint32_t) * ((size_t)x.img_ncol + 1u) * g.ncell, st);
    if (x.merge) hipLaunchKernelGGL(k_wire_img_need, dim3((w.S + 3) / 4), dim3(256), 0, st, w, x);
    hipLaunchKernelGGL(k_wire_img_sizes, dim3(g.ncell, ncol + 1u), dim3(256), 0, st, g, w, x, ncol);
    launch_scan_u32_inplace(st, x.img_off[0], ncol * g.ncell);
    launch_scan_u32_inplace(st, x.img_off[1], g.ncell);
}

// ... and, after the sizing pass of the layout has marked them (img_need), the bytes of the images the tick copies from
void launch_wire_images_fill(hipStream_t st, DevGrid g, WorldDev w, WireDev x) {
    if (!g.ncell) return;
    const uint32_t ncol = wire_image_columns(w, x);
    hipLaunchKernelGGL(k_wire_img_fill, dim3(g.ncell, (ncol + 1u) * WIRE_FILL_SPLIT), dim3(256), 0, st, g, w, x, ncol);
}

void launch_wire_layout_img(hipStream_t st, DevGrid g, WorldDev w, WireDev x, bool emit) {
    if (!w.S) return;
    if (emit) hipLaunchKernelGGL(k_wire_layout_img<true>, dim3((w.S + 3) / 4), dim3(256), 0, st, g, w, x);
    else hipLaunchKernelGGL(k_wire_layout_img<false>, dim3((w.S + 3) / 4), dim3(256), 0, st, g, w, x);
}

void launch_wire_copy_img(hipStream_t st, WorldDev w, WireDev x, uint32_t waves) {
    static const uint32_t mult = [] { const char *e = getenv("CHD_WIRE_COPY_WAVES"); return e ? (uint32_t)atoi(e) : 2u; }();  // x 8 waves per CU
    waves *= mult ? mult : 1u;
    static const bool nt = [] { const char *e = getenv("CHD_WIRE_COPY_NT"); return e && e[0] == '1'; }();
    // (everything sized on the device: the descriptor count and the list of record-path connections are not known to the host yet)
    hipLaunchKernelGGL(k_wire_copy_ends, dim3(16384), dim3(256), 0, st, w, x);
    const uint32_t wgs = waves / 4u ? waves / 4u : 1u;  // (four waves each)
    if (nt) hipLaunchKernelGGL(k_wire_copy_img<true>, dim3(wgs), dim3(256), 0, st, w, x);
    else hipLaunchKernelGGL(k_wire_copy_img<false>, dim3(wgs), dim3(256), 0, st, w, x);
    hipLaunchKernelGGL(k_wire_copy_list, dim3(512), dim3(256), 0, st, w, x);
}

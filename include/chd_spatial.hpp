// chd_spatial.hpp — C++17 host-side mirror of channeld's Go `SpatialController` interface
// (pkg/channeld/spatial.go:17-35) over the C-ABI of libchd_spatial.so (chd_spatial.h).  Header-only.
//
// channeld is Go and this image has no Go toolchain, so the host side above the C-ABI exists twice: this header
// (the reference is compiled code) and the Python mirror channeld_amd/controller.py + engine.py that the test-suite
// drives.  Same method names, argument meaning and error behaviour as the Go type StaticGrid2DSpatialController
// (spatial.go:89-124): methods return {value, Error} where Go returns (value, error); Error is "nil" when it
// converts to false.  All batched arithmetic runs in HIP kernels behind the C-ABI — nothing here computes cells,
// AOIs or fan-out; the one piece of arithmetic is go_cos (the cone's cos(angle), which a Go caller would take from
// Go's own math.Cos, spatial.go:295).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "chd_spatial.h"

namespace chd {

using ChannelId = uint32_t;    // common.ChannelId
using ConnectionId = uint32_t;

constexpr ChannelId SpatialChannelIdStart = 0x10000;  // settings.go:94
constexpr ChannelId EntityChannelIdStart = 0x80000;   // settings.go:95
constexpr double MinY = -3.40282347e38 / 2;           // spatial.go:80-83
constexpr double MaxY = 3.40282347e38 / 2;

// The `error` half of a Go (value, error) return: false == nil.
struct Error {
    int code = CHD_OK;
    std::string msg;
    explicit operator bool() const { return code != CHD_OK; }
};

struct SpatialInfo {  // pkg/common/common.go:20-24
    double X = 0, Y = 0, Z = 0;
};
struct SpotsAOI {  // channeld.proto SpatialInterestQuery.SpotsAOI
    std::vector<SpatialInfo> Spots;
    std::vector<uint32_t> Dists;
};
struct BoxAOI {
    std::optional<SpatialInfo> Center, Extent;
};
struct SphereAOI {
    std::optional<SpatialInfo> Center;
    double Radius = 0;
};
struct ConeAOI {
    std::optional<SpatialInfo> Center, Direction;
    double Angle = 0, Radius = 0;
};
struct SpatialInterestQuery {  // channeld.proto:386-440; an absent shape is a nil sub-message
    std::optional<chd::SpotsAOI> SpotsAOI;
    std::optional<chd::BoxAOI> BoxAOI;
    std::optional<chd::SphereAOI> SphereAOI;
    std::optional<chd::ConeAOI> ConeAOI;
};
struct SpatialRegion {  // channeldpb.SpatialRegion
    SpatialInfo Min, Max;
    ChannelId ChannelId_ = 0;
    uint32_t ServerIndex = 0;
};

// Go's math.Cos (Go standard library src/math/sin.go, the Cephes port) for |x| < 2^29, every operation rounded on its
// own (compile the including file without -ffast-math; volatile keeps the products from fusing into FMAs).  The same
// restatement as channeld_amd/gomath.py and the oracle's orc_go_cos: written from the published algorithm, compared
// with each other bit for bit (tests/test_cxx_host.py), NOT checked against a Go toolchain.
inline double go_cos(double x) {
    if (std::isnan(x) || std::isinf(x)) return std::nan("");
    static const double PI4A = 7.85398125648498535156e-1, PI4B = 3.77489470793079817668e-8, PI4C = 2.69515142907905952645e-15;
    static const double S[6] = {1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
                                -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1};
    static const double C[6] = {-1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
                                2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2};
    static const double M4PI = 1.2732395447351628;
    bool sign = false;
    x = std::fabs(x);
    if (x >= (double)(1 << 29)) throw std::domain_error("go_cos: Payne-Hanek range (|x| >= 2^29) is not restated");
    volatile double t = x * M4PI;
    int64_t j = (int64_t)t;
    double y = (double)j;
    if (j & 1) { j += 1; y += 1.0; }
    j &= 7;
    volatile double a = y * PI4A, b = y * PI4B, c = y * PI4C;
    volatile double z1 = x - a;
    volatile double z2 = z1 - b;
    volatile double z = z2 - c;
    if (j > 3) { j -= 4; sign = !sign; }
    if (j > 1) sign = !sign;
    volatile double zz = z * z;
    const double *P = (j == 1 || j == 2) ? S : C;
    volatile double p = P[0] * zz;
    for (int k = 1; k < 6; k++) {
        volatile double s = p + P[k];
        p = (k < 5) ? s * zz : s;
    }
    volatile double r;
    if (j == 1 || j == 2) {
        volatile double zzz = z * zz;
        volatile double q = zzz * p;
        r = z + q;
    } else {
        volatile double h = 0.5 * zz;
        volatile double one = 1.0 - h;
        volatile double z4 = zz * zz;
        volatile double q = z4 * p;
        r = one + q;
    }
    return sign ? -r : r;
}

// SpatialInterestQuery messages -> chd_aoi_query records + the spot side arrays (as controller.py: pack_queries).
// A nil Center / Extent / Direction would nil-deref in Go (spatial.go:205,237,272): reported as CHD_E_INVAL.
struct PackedQueries {
    std::vector<chd_aoi_query> q;
    std::vector<double> spot_x, spot_z;
    std::vector<uint32_t> spot_dist;
};
inline Error pack_queries(const std::vector<const SpatialInterestQuery *> &queries, PackedQueries &out) {
    out = PackedQueries{};
    out.q.resize(queries.size());
    for (size_t i = 0; i < queries.size(); i++) {
        const SpatialInterestQuery *q = queries[i];
        if (!q) return {CHD_E_INVAL, "query is nil"};  // spatial.go:183-185
        chd_aoi_query &a = out.q[i];
        std::memset(&a, 0, sizeof a);
        if (q->SpotsAOI) {
            a.shapes |= CHD_SHAPE_SPOTS;
            a.spot_off = (uint32_t)out.spot_x.size();
            a.n_spots = (uint32_t)q->SpotsAOI->Spots.size();
            a.n_spot_dists = (uint32_t)std::min(q->SpotsAOI->Dists.size(), q->SpotsAOI->Spots.size());
            for (size_t k = 0; k < q->SpotsAOI->Spots.size(); k++) {
                out.spot_x.push_back(q->SpotsAOI->Spots[k].X);
                out.spot_z.push_back(q->SpotsAOI->Spots[k].Z);
                out.spot_dist.push_back(k < q->SpotsAOI->Dists.size() ? q->SpotsAOI->Dists[k] : 0u);
            }
        }
        if (q->BoxAOI) {
            if (!q->BoxAOI->Center || !q->BoxAOI->Extent) return {CHD_E_INVAL, "BoxAOI.Center/Extent is nil"};
            a.shapes |= CHD_SHAPE_BOX;
            a.box_cx = q->BoxAOI->Center->X; a.box_cz = q->BoxAOI->Center->Z;
            a.box_ex = q->BoxAOI->Extent->X; a.box_ez = q->BoxAOI->Extent->Z;
        }
        if (q->SphereAOI) {
            if (!q->SphereAOI->Center) return {CHD_E_INVAL, "SphereAOI.Center is nil"};
            a.shapes |= CHD_SHAPE_SPHERE;
            a.sph_cx = q->SphereAOI->Center->X; a.sph_cz = q->SphereAOI->Center->Z; a.sph_r = q->SphereAOI->Radius;
        }
        if (q->ConeAOI) {
            if (!q->ConeAOI->Center || !q->ConeAOI->Direction) return {CHD_E_INVAL, "ConeAOI.Center/Direction is nil"};
            a.shapes |= CHD_SHAPE_CONE;
            a.cone_cx = q->ConeAOI->Center->X; a.cone_cz = q->ConeAOI->Center->Z;
            a.cone_dx = q->ConeAOI->Direction->X; a.cone_dz = q->ConeAOI->Direction->Z;
            a.cone_r = q->ConeAOI->Radius;
            a.cone_cos = go_cos(q->ConeAOI->Angle);  // math.Cos(Angle), spatial.go:295
        }
    }
    return {};
}

namespace detail {
// The flat JSON object LoadConfig unmarshals (spatial.go:141-145): "Key": number pairs, optionally wrapped in
// {"SpatialControllerType": ..., "Config": {...}} as InitSpatialController reads the file (:61-68).
inline bool json_number(std::string_view js, const char *key, double &out, bool &present) {
    present = false;
    const std::string pat = std::string("\"") + key + "\"";
    size_t p = js.find(pat);
    if (p == std::string_view::npos) return true;
    p = js.find(':', p + pat.size());
    if (p == std::string_view::npos) return false;
    p++;
    while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) p++;
    const std::string num(js.substr(p, std::min<size_t>(64, js.size() - p)));
    char *end = nullptr;
    const double v = std::strtod(num.c_str(), &end);
    if (end == num.c_str()) return false;  // not a number: json.Unmarshal fails
    out = v;
    present = true;
    return true;
}
}  // namespace detail

// settings the reference takes from GlobalSettings (settings.go:64-105) and message_spatial.go:16-29
struct ControllerSettings {
    ChannelId SpatialChannelIdStart = chd::SpatialChannelIdStart, EntityChannelIdStart = chd::EntityChannelIdStart;
    uint32_t DefaultFanOutIntervalMs = 20;
    int32_t DefaultFanOutDelayMs = 0;
    std::vector<std::pair<uint32_t, uint32_t>> Damping;  // {MaxDistance, FanOutIntervalMs}; empty = the reference's table
};

// Second implementation of `SpatialController` (the first being the reference's Go type of the same name), backed by
// the gfx950 library.  The nine exported fields are the reference's (spatial.go:89-124).
class StaticGrid2DSpatialController {
  public:
    double GridWidth = 0, GridHeight = 0, WorldOffsetX = 0, WorldOffsetZ = 0;
    uint32_t GridCols = 0, GridRows = 0, ServerCols = 0, ServerRows = 0, ServerInterestBorderSize = 0;

    using Settings = ControllerSettings;

    explicit StaticGrid2DSpatialController(int device = 0) : device_(device) {}
    StaticGrid2DSpatialController(const StaticGrid2DSpatialController &) = delete;
    StaticGrid2DSpatialController &operator=(const StaticGrid2DSpatialController &) = delete;
    ~StaticGrid2DSpatialController() { close(); }
    void close() {
        if (ctx_) chd_destroy(ctx_);
        ctx_ = nullptr;
    }
    chd_ctx *ctx() const { return ctx_; }

    // spatial.go:141-159.  strict = LoadConfig's own rejection of ServerInterestBorderSize <= 0 (:155), which
    // InitSpatialController ignores (:68): pass false for that path.
    Error LoadConfig(std::string_view config, bool strict = true, const Settings &st = Settings()) {
        std::string_view js = config;
        const size_t c = js.find("\"Config\"");
        if (c != std::string_view::npos) js = js.substr(c + 8);
        chd_grid_cfg cfg;
        std::memset(&cfg, 0, sizeof cfg);
        struct F { const char *k; double *d; uint32_t *u; } fields[] = {
            {"GridWidth", &cfg.grid_width, nullptr}, {"GridHeight", &cfg.grid_height, nullptr},
            {"WorldOffsetX", &cfg.world_offset_x, nullptr}, {"WorldOffsetZ", &cfg.world_offset_z, nullptr},
            {"GridCols", nullptr, &cfg.grid_cols}, {"GridRows", nullptr, &cfg.grid_rows},
            {"ServerCols", nullptr, &cfg.server_cols}, {"ServerRows", nullptr, &cfg.server_rows},
            {"ServerInterestBorderSize", nullptr, &cfg.server_interest_border_size}};
        for (const F &f : fields) {
            double v = 0;
            bool present = false;
            if (!detail::json_number(js, f.k, v, present)) return {CHD_E_CONFIG, std::string("json: cannot unmarshal field ") + f.k};
            if (!present) continue;
            if (f.d) *f.d = v;
            else {
                if (v < 0 || v != std::floor(v) || v > 4294967295.0)
                    return {CHD_E_CONFIG, std::string("json: cannot unmarshal number into uint32 field ") + f.k};
                *f.u = (uint32_t)v;
            }
        }
        cfg.spatial_channel_id_start = st.SpatialChannelIdStart;
        cfg.entity_channel_id_start = st.EntityChannelIdStart;
        cfg.default_fanout_interval_ms = st.DefaultFanOutIntervalMs;
        cfg.default_fanout_delay_ms = st.DefaultFanOutDelayMs;
        if (st.Damping.size() > CHD_MAX_DAMPING) return {CHD_E_CONFIG, "damping table longer than CHD_MAX_DAMPING"};
        cfg.n_damping = (uint32_t)st.Damping.size();
        for (size_t i = 0; i < st.Damping.size(); i++) {
            cfg.damping_max_dist[i] = st.Damping[i].first;
            cfg.damping_interval_ms[i] = st.Damping[i].second;
        }
        cfg.strict_load_config = strict ? 1u : 0u;
        close();
        chd_ctx *ctx = nullptr;
        const int rc = chd_create(&cfg, device_, &ctx);
        if (rc != CHD_OK) {
            const char *m = chd_last_error(nullptr);
            return {rc, m ? m : "chd_create failed"};  // CHD_E_CONFIG: the reference's validation; CHD_E_NO_DEVICE: no gfx950, no fallback
        }
        ctx_ = ctx;
        GridWidth = cfg.grid_width; GridHeight = cfg.grid_height;
        WorldOffsetX = cfg.world_offset_x; WorldOffsetZ = cfg.world_offset_z;
        GridCols = cfg.grid_cols; GridRows = cfg.grid_rows;
        ServerCols = cfg.server_cols; ServerRows = cfg.server_rows;
        ServerInterestBorderSize = cfg.server_interest_border_size;
        spatial_id_start_ = cfg.spatial_channel_id_start;
        serverConnections_.assign((size_t)ServerCols * ServerRows, 0);
        return {};
    }

    // spatial.go:161-163: (id, nil) or (0, err)
    std::pair<ChannelId, Error> GetChannelId(const SpatialInfo &info) {
        uint32_t id = 0;
        const int rc = chd_get_channel_ids(need(), &info.X, &info.Z, 1, &id);
        if (rc != CHD_OK) return {0, last(rc)};
        if (id == 0) return {0, {CHD_E_INVAL, "spatial info is outside the grid"}};
        return {id, {}};
    }
    // batched form: 0 = out of the world
    Error GetChannelIds(const std::vector<double> &x, const std::vector<double> &z, std::vector<ChannelId> &out) {
        out.assign(x.size(), 0);
        if (x.size() != z.size()) return {CHD_E_INVAL, "x and z differ in length"};
        if (x.empty()) return {};
        const int rc = chd_get_channel_ids(need(), x.data(), z.data(), (uint32_t)x.size(), out.data());
        return rc == CHD_OK ? Error{} : last(rc);
    }

    // spatial.go:182-317: (map[ChannelId]uint, nil) or (nil, err)
    std::pair<std::map<ChannelId, uint32_t>, Error> QueryChannelIds(const SpatialInterestQuery *query) {
        std::map<ChannelId, uint32_t> result;
        PackedQueries p;
        if (Error e = pack_queries({query}, p)) return {result, e};
        const uint32_t cap = std::max<uint32_t>(1u, std::min<uint32_t>(GridCols * GridRows, 4096u));
        std::vector<uint32_t> ids(cap), dists(cap);
        uint32_t off[2] = {0, 0};
        int32_t status = CHD_OK;
        const int rc = chd_query_channel_ids(need(), p.q.data(), 1, p.spot_x.empty() ? nullptr : p.spot_x.data(),
                                             p.spot_z.empty() ? nullptr : p.spot_z.data(),
                                             p.spot_dist.empty() ? nullptr : p.spot_dist.data(), (uint32_t)p.spot_x.size(), off,
                                             ids.data(), dists.data(), nullptr, cap, &status);
        if (rc != CHD_OK) return {result, last(rc)};
        if (status != CHD_OK) return {result, {status, query_error_text(status)}};  // the reference's error returns (:208-215, :228-231, ...)
        for (uint32_t i = off[0]; i < off[1]; i++) result[ids[i]] = dists[i];
        return {result, {}};
    }

    // spatial.go:319-356
    std::pair<std::vector<SpatialRegion>, Error> GetRegions() {
        const size_t n = (size_t)GridCols * GridRows;
        std::vector<double> a(n), b(n), c(n), d(n);
        std::vector<uint32_t> cid(n), srv(n);
        std::vector<SpatialRegion> out;
        const int rc = chd_get_regions(need(), a.data(), b.data(), c.data(), d.data(), cid.data(), srv.data());
        if (rc != CHD_OK) return {out, last(rc)};
        out.resize(n);
        for (size_t i = 0; i < n; i++) out[i] = SpatialRegion{{a[i], MinY, b[i]}, {c[i], MaxY, d[i]}, cid[i], srv[i]};
        return {out, {}};
    }

    // spatial.go:358-381: the up-to-8 neighbours in the reference's row-major order
    std::pair<std::vector<ChannelId>, Error> GetAdjacentChannels(ChannelId spatialChannelId) {
        uint32_t out[8], cnt = 0;
        const int rc = chd_get_adjacent_channels(need(), &spatialChannelId, 1, out, &cnt);
        if (rc != CHD_OK) return {{}, last(rc)};
        return {std::vector<ChannelId>(out, out + cnt), {}};
    }

    // The spatial part of CreateChannels (spatial.go:387-479): gives `connection` the next server slot and returns the
    // spatial channel ids it owns; once the last server has arrived `borderChannels(i)` are what server i is subscribed
    // to (subToAdjacentChannels, :481-590).  Channel objects and messages stay with the gateway.
    std::pair<std::vector<ChannelId>, Error> CreateChannels(ConnectionId connection) {
        const uint32_t idx = nextServerIndex(), total = ServerCols * ServerRows;
        if (idx >= total) return {{}, {CHD_E_INVAL, "all grids are allocated to the servers"}};
        auto r = ServerChannels(idx);
        if (r.second) return r;
        serverConnections_[idx] = connection;
        return r;
    }
    std::pair<std::vector<ChannelId>, Error> ServerChannels(uint32_t serverIndex) { return list_call(chd_server_channels, serverIndex, GridCols * GridRows + 8); }
    std::pair<std::vector<ChannelId>, Error> BorderChannels(uint32_t serverIndex) {
        return list_call(chd_border_channels, serverIndex, 4 * (GridCols + GridRows) * std::max(ServerInterestBorderSize, 1u) + 8);
    }
    uint32_t nextServerIndex() const {  // spatial.go:866-874
        for (size_t i = 0; i < serverConnections_.size(); i++)
            if (serverConnections_[i] == 0) return (uint32_t)i;
        return (uint32_t)serverConnections_.size();
    }
    void ServerConnectionClosed(ConnectionId connection) {  // what Tick() finds out through IsClosing(), spatial.go:876-884
        for (auto &c : serverConnections_)
            if (c == connection) c = 0;
    }

    // spatial.go:612-626, the decision: calls the provider with (src, dst) iff both positions are in the world and in
    // different cells.  Everything after it is the gateway's; the batched engine is SpatialWorld::Tick.
    void Notify(const SpatialInfo &oldInfo, const SpatialInfo &newInfo,
                const std::function<void(ChannelId, ChannelId, void *)> &handoverDataProvider) {
        uint32_t src = 0, dst = 0;
        uint8_t ho = 0;
        if (chd_notify_decide(need(), &oldInfo.X, &oldInfo.Z, &newInfo.X, &newInfo.Z, 1, &src, &dst, &ho) != CHD_OK) return;
        if (ho) handoverDataProvider(src, dst, nullptr);
    }

    static const char *query_error_text(int status) {
        switch (status) {
            case CHD_E_EXTENT: return "invalid box extent / radius";
            case CHD_E_CENTER: return "AOI centre is outside the world";
            case CHD_E_CAPACITY: return "interest set exceeds the engine capacity";
            case CHD_E_HANG: return "the reference implementation would not terminate on this query";
            case CHD_E_TOO_LARGE: return "sample lattice or cell window beyond engine limits";
            default: return "query failed";
        }
    }

  private:
    chd_ctx *need() const {
        if (!ctx_) throw std::logic_error("LoadConfig has not been called");
        return ctx_;
    }
    Error last(int rc) const {
        const char *m = chd_last_error(ctx_);
        return {rc, m ? m : ""};
    }
    template <typename Fn>
    std::pair<std::vector<ChannelId>, Error> list_call(Fn fn, uint32_t serverIndex, uint32_t cap) {
        std::vector<ChannelId> out(cap);
        uint32_t n = 0;
        const int rc = fn(need(), serverIndex, out.data(), cap, &n);
        if (rc != CHD_OK) return {{}, last(rc)};
        out.resize(n);
        return {out, {}};
    }
    int device_;
    chd_ctx *ctx_ = nullptr;
    ChannelId spatial_id_start_ = chd::SpatialChannelIdStart;
    std::vector<ConnectionId> serverConnections_;  // 0 = free slot
};

// What the per-message callers of the path become on the host side of the boundary (a cgo / FFI call per message is not
// affordable: the shim buffers and resolves at the next tick).  Between two ticks every entity-channel update message —
// Channel.tickMessages -> ChannelData.OnUpdate + Notify (channel.go:296-310, data.go:149-173, spatial.go:612) —, every update
// of a spatial channel's own data and every UPDATE_SPATIAL_INTEREST (message_spatial.go:59) is recorded here in ARRIVAL order;
// Layout() arranges them as chd_tick_in wants them:
//   exact worlds (history_depth > 0)  every update with its own arrival stamp; a channel's r-th update of the tick goes to round
//                                     r (upd_round_off: one update per entity per round, rounds applied in order = the
//                                     reference's message order per channel)
//   ring worlds                       one update per entity and tick: the LAST one (position and sender), stamped by the tick
//   interest                          one update per connection and tick: the last one
// (channeld_amd/engine.py: UpdateBatch is the same, array for array — tests/test_cxx_host.py.)
class UpdateBatch {
  public:
    explicit UpdateBatch(bool exact) : exact_(exact) {}
    void OnUpdate(uint32_t slot, double x, double z, ConnectionId sender, int64_t arrivalNs) {
        slot_.push_back(slot); x_.push_back(x); z_.push_back(z); sender_.push_back(sender); arrival_.push_back(arrivalNs);
    }
    void OnCellUpdate(ChannelId channel, ConnectionId sender, int64_t arrivalNs) {
        cellChannel.push_back(channel); cellSender.push_back(sender); cellArrivalNs.push_back(arrivalNs);
    }
    void OnInterest(uint32_t subSlot, const SpatialInterestQuery *query) { interest_.push_back({subSlot, query}); }
    void Clear() { *this = UpdateBatch(exact_); }
    bool Exact() const { return exact_; }

    // the arrays of chd_tick_in (valid until the next OnUpdate / Clear)
    std::vector<uint32_t> updSlot, updSender, roundOff;  // roundOff: empty on ring worlds
    std::vector<double> updX, updZ;
    std::vector<int64_t> updArrivalNs;                   // empty on ring worlds
    std::vector<ChannelId> cellChannel;
    std::vector<ConnectionId> cellSender;
    std::vector<int64_t> cellArrivalNs;
    std::vector<uint32_t> querySub;
    std::vector<const SpatialInterestQuery *> queries;

    void Layout() {
        const size_t n = slot_.size();
        std::vector<size_t> order;
        roundOff.clear();
        if (!exact_) {
            // the last update of every entity, entities in the order of their last update
            std::map<uint32_t, size_t> last;
            for (size_t i = 0; i < n; i++) last[slot_[i]] = i;
            for (auto &kv : last) order.push_back(kv.second);
            std::sort(order.begin(), order.end());
        } else if (n) {
            std::map<uint32_t, uint32_t> seen;
            std::vector<uint32_t> rnd(n);
            uint32_t rounds = 0;
            for (size_t i = 0; i < n; i++) { rnd[i] = seen[slot_[i]]++; rounds = std::max(rounds, rnd[i] + 1); }
            roundOff.assign((size_t)rounds + 1, 0);
            for (size_t i = 0; i < n; i++) roundOff[rnd[i] + 1]++;
            for (uint32_t r = 0; r < rounds; r++) roundOff[r + 1] += roundOff[r];
            order.resize(n);
            std::vector<uint32_t> at(roundOff.begin(), roundOff.end() - 1);
            for (size_t i = 0; i < n; i++) order[at[rnd[i]]++] = i;  // round-major, arrival order inside a round
        }
        updSlot.clear(); updSender.clear(); updX.clear(); updZ.clear(); updArrivalNs.clear();
        for (size_t i : order) {
            updSlot.push_back(slot_[i]); updSender.push_back(sender_[i]); updX.push_back(x_[i]); updZ.push_back(z_[i]);
            if (exact_) updArrivalNs.push_back(arrival_[i]);
        }
        // one interest update per connection: the last one, connections in the order of their last query
        querySub.clear(); queries.clear();
        std::map<uint32_t, size_t> lastq;
        for (size_t i = 0; i < interest_.size(); i++) lastq[interest_[i].first] = i;
        std::vector<size_t> qi;
        for (auto &kv : lastq) qi.push_back(kv.second);
        std::sort(qi.begin(), qi.end());
        for (size_t i : qi) { querySub.push_back(interest_[i].first); queries.push_back(interest_[i].second); }
    }

  private:
    bool exact_;
    std::vector<uint32_t> slot_, sender_;
    std::vector<double> x_, z_;
    std::vector<int64_t> arrival_;
    std::vector<std::pair<uint32_t, const SpatialInterestQuery *>> interest_;
};

// The batched engine: one Tick replaces N Notify calls, S handleUpdateSpatialInterest calls and tickData on every
// spatial and entity channel (chd_tick).  Buffers are owned here and reused from tick to tick.
class SpatialWorld {
  public:
    struct TickResult {
        std::vector<chd_handover_rec> handovers;
        uint32_t lockedAborts = 0;
        std::vector<int32_t> queryStatus;
        std::vector<uint32_t> unsubSlot, unsubChannel, newSubSlot, newSubChannel, newSubIntervalMs;
        std::vector<chd_fanout_rec> records;       // grouped per connection slot:
        std::vector<uint64_t> connRecordOffset;    //   slot s owns records [connRecordOffset[s], + connRecordCount[s])
        std::vector<uint32_t> connRecordCount;
        // what crossed PCIe (chd_tick_fetch_segments): `records` above is expanded from these on the host
        std::vector<chd_fanout_segment> segments;  // slot s owns segments [connSegmentOffset[s], connSegmentOffset[s+1])
        std::vector<uint32_t> connSegmentOffset, columns;
        std::vector<chd_fanout_rec> explicitRecords;
        std::vector<uint64_t> connExplicitOffset;
        uint32_t overflow = 0, historyOverflow = 0;
    };

    SpatialWorld(StaticGrid2DSpatialController &ctl, uint32_t maxEntities, uint32_t maxSubscribers, uint64_t maxRecords = 0,
                 uint32_t flags = 0)
        : ctl_(ctl), N_(maxEntities), S_(maxSubscribers) {
        chd_world_cfg c;
        std::memset(&c, 0, sizeof c);
        c.max_entities = maxEntities;
        c.max_subscribers = maxSubscribers;
        c.max_records = maxRecords;
        c.flags = flags;
        check(chd_world_create(ctl.ctx(), &c));
        capq_ = std::min<uint32_t>(ctl.GridCols * ctl.GridRows, 256u);
    }
    // entity channel creation + spawn (message_spatial.go:231-237): slots 0..n-1
    void Spawn(const std::vector<ChannelId> &entityChannelIds, const std::vector<double> &x, const std::vector<double> &z,
               const std::vector<uint32_t> &flags, const std::vector<ConnectionId> &owner) {
        check(chd_world_spawn(ctl_.ctx(), (uint32_t)entityChannelIds.size(), nullptr, entityChannelIds.data(), x.data(), z.data(),
                              flags.empty() ? nullptr : flags.data(), owner.empty() ? nullptr : owner.data()));
    }
    void AddSubscribers(const std::vector<ConnectionId> &conn) {
        check(chd_subs_add(ctl_.ctx(), (uint32_t)conn.size(), nullptr, conn.data()));
        if (connIds_.size() < conn.size()) connIds_.resize(conn.size(), 0);
        std::copy(conn.begin(), conn.end(), connIds_.begin());  // (slots 0 .. n-1)
    }

    // The fan-out of the last tick as the host consumes it: segment descriptors + the cells' entity-channel columns + the few
    // explicit records (chd_tick_fetch_segments), expanded here into the per-connection records a flush loop would walk —
    // what Go's fanOutDataUpdate loop (data.go:293-318) becomes on the host side of the boundary.
    Error FetchSegments(TickResult &out) {
        out.connSegmentOffset.assign((size_t)S_ + 1, 0);
        out.connExplicitOffset.assign((size_t)S_ + 1, 0);
        chd_segments_out so;
        std::memset(&so, 0, sizeof so);
        for (int attempt = 0; attempt < 2; attempt++) {
            so.segments = out.segments.data(); so.segments_cap = out.segments.size();
            so.conn_seg_off = out.connSegmentOffset.data();
            so.columns = out.columns.data(); so.columns_cap = out.columns.size();
            so.records = out.explicitRecords.data(); so.records_cap = out.explicitRecords.size();
            so.conn_rec_off = out.connExplicitOffset.data();
            const int rc = chd_tick_fetch_segments(ctl_.ctx(), &so);
            if (rc == CHD_E_CAPACITY && attempt == 0 && (so.n_segments > so.segments_cap || so.n_columns > so.columns_cap || so.n_explicit > so.records_cap)) {
                out.segments.resize(so.n_segments + so.n_segments / 4 + 16);
                out.columns.resize(so.n_columns + 16);
                out.explicitRecords.resize(so.n_explicit + so.n_explicit / 4 + 16);
                continue;
            }
            if (rc != CHD_OK) return err(rc);
            break;
        }
        out.segments.resize(so.n_segments);
        out.explicitRecords.resize(so.n_explicit);
        // expansion (include/chd_spatial.h: chd_fanout_segment)
        out.records.clear();
        out.records.reserve(so.n_records);
        out.connRecordOffset.assign((size_t)S_ + 1, 0);
        out.connRecordCount.assign(S_, 0);
        for (uint32_t s = 0; s < S_; s++) {
            out.connRecordOffset[s] = out.records.size();
            const uint32_t conn = s < connIds_.size() ? connIds_[s] : 0u;
            for (uint32_t k = out.connSegmentOffset[s]; k < out.connSegmentOffset[s + 1]; k++) {
                const chd_fanout_segment &g = out.segments[k];
                if (g.n_info & CHD_SEG_EXPLICIT) {
                    const chd_fanout_rec *r = out.explicitRecords.data() + out.connExplicitOffset[s] + g.off;
                    out.records.insert(out.records.end(), r, r + g.n_records);
                    continue;
                }
                const uint32_t n = CHD_SEG_N(g.n_info);
                const uint32_t *col = out.columns.data() + g.off;
                if (g.n_info & CHD_SEG_FIRST) {
                    out.records.push_back({conn | CHD_REC_FULL, g.channel});
                    for (uint32_t e = 0; e < n; e++) out.records.push_back({conn | CHD_REC_FULL, col[e]});
                }
                for (uint32_t j = 0; j < CHD_SEG_NWIN(g.n_info); j++) {
                    if (g.n_info & CHD_SEG_OWN(j)) out.records.push_back({conn, g.channel});
                    if (!(g.n_info & CHD_SEG_NONE))
                        for (uint32_t e = 0; e < n; e++) out.records.push_back({conn, col[e]});
                }
            }
            out.connRecordCount[s] = (uint32_t)(out.records.size() - out.connRecordOffset[s]);
        }
        out.connRecordOffset[S_] = out.records.size();
        if (out.records.size() != so.n_records) return {CHD_E_STATE, "segments expand to another number of records than the tick reported"};
        return {};
    }

    // entity updates (slot u = u), interest updates (one query per connection slot, nullptr = none), fan-out at nowNs
    Error Tick(int64_t nowNs, const std::vector<double> &x, const std::vector<double> &z,
               const std::vector<const SpatialInterestQuery *> &queries, uint64_t recordsCap, TickResult &out) {
        PackedQueries p;
        if (Error e = pack_queries(queries, p)) return e;
        chd_tick_in in;
        std::memset(&in, 0, sizeof in);
        in.now_ns = nowNs;
        in.n_updates = (uint32_t)x.size();
        in.upd_x = x.data();
        in.upd_z = z.data();
        in.n_queries = (uint32_t)p.q.size();
        in.queries = p.q.empty() ? nullptr : p.q.data();
        in.spot_x = p.spot_x.empty() ? nullptr : p.spot_x.data();
        in.spot_z = p.spot_z.empty() ? nullptr : p.spot_z.data();
        in.spot_dist = p.spot_dist.empty() ? nullptr : p.spot_dist.data();
        in.n_spots_total = (uint32_t)p.spot_x.size();
        return RunTick(in, p.q.size(), recordsCap, out);
    }

  private:
    Error RunTick(const chd_tick_in &in, size_t nQueries, uint64_t recordsCap, TickResult &out) {
        const uint32_t listCap = std::max<uint32_t>(1u, S_ * capq_);  // (every connection can drop / gain a whole interest set)
        out.handovers.resize(std::max<uint32_t>(N_, 1u));
        out.queryStatus.assign(std::max<size_t>(nQueries, 1), 0);
        for (auto *v : {&out.unsubSlot, &out.unsubChannel, &out.newSubSlot, &out.newSubChannel, &out.newSubIntervalMs}) v->resize(listCap);
        chd_tick_out o;
        std::memset(&o, 0, sizeof o);
        o.handovers = out.handovers.data(); o.handovers_cap = (uint32_t)out.handovers.size();
        o.query_status = out.queryStatus.data();
        o.unsub_sub = out.unsubSlot.data(); o.unsub_channel = out.unsubChannel.data(); o.unsub_cap = listCap;
        o.newsub_sub = out.newSubSlot.data(); o.newsub_channel = out.newSubChannel.data(); o.newsub_interval_ms = out.newSubIntervalMs.data();
        o.newsub_cap = listCap;
        // (no dense records over PCIe: the fan-out comes back as segments, below)
        const int rc = chd_tick(ctl_.ctx(), &in, &o);
        out.overflow = o.overflow;
        out.historyOverflow = o.history_overflow;
        if (rc != CHD_OK) {
            const char *m = chd_last_error(ctl_.ctx());
            return {rc, m ? m : ""};
        }
        out.handovers.resize(o.n_handovers);
        out.lockedAborts = o.n_locked_aborts;
        for (auto *v : {&out.unsubSlot, &out.unsubChannel}) v->resize(o.n_unsubs);
        for (auto *v : {&out.newSubSlot, &out.newSubChannel, &out.newSubIntervalMs}) v->resize(o.n_newsubs);
        if (o.n_records > recordsCap) return {CHD_E_CAPACITY, "more fan-out records than recordsCap"};
        if (Error e = FetchSegments(out)) return e;
        return {};
    }

  public:

    // the same tick from the messages recorded since the last one (UpdateBatch: any subset of the entities, several updates
    // per entity and their arrival stamps on exact worlds, spatial-channel updates, interest updates of some connections)
    Error Tick(int64_t nowNs, UpdateBatch &batch, uint64_t recordsCap, TickResult &out) {
        batch.Layout();
        PackedQueries p;
        if (Error e = pack_queries(batch.queries, p)) return e;
        chd_tick_in in;
        std::memset(&in, 0, sizeof in);
        in.now_ns = nowNs;
        in.n_updates = (uint32_t)batch.updSlot.size();
        in.upd_idx = batch.updSlot.data();
        in.upd_x = batch.updX.data();
        in.upd_z = batch.updZ.data();
        in.upd_sender = batch.updSender.data();
        if (batch.Exact() && in.n_updates) {
            in.upd_arrival_ns = batch.updArrivalNs.data();
            in.n_update_rounds = (uint32_t)batch.roundOff.size() - 1u;
            in.upd_round_off = batch.roundOff.data();
        }
        in.n_cell_updates = (uint32_t)batch.cellChannel.size();
        in.cell_upd_channel = batch.cellChannel.empty() ? nullptr : batch.cellChannel.data();
        in.cell_upd_sender = batch.cellSender.empty() ? nullptr : batch.cellSender.data();
        if (batch.Exact() && in.n_cell_updates) in.cell_upd_arrival_ns = batch.cellArrivalNs.data();
        in.n_queries = (uint32_t)p.q.size();
        in.query_sub = batch.querySub.empty() ? nullptr : batch.querySub.data();
        in.queries = p.q.empty() ? nullptr : p.q.data();
        in.spot_x = p.spot_x.empty() ? nullptr : p.spot_x.data();
        in.spot_z = p.spot_z.empty() ? nullptr : p.spot_z.data();
        in.spot_dist = p.spot_dist.empty() ? nullptr : p.spot_dist.data();
        in.n_spots_total = (uint32_t)p.spot_x.size();
        const Error e = RunTick(in, p.q.size(), recordsCap, out);
        if (!e) batch.Clear();
        return e;
    }

    // ---- the rest of the world API, one thin method per entry point (chd_spatial.h has the reference citations) ----
    void Despawn(const std::vector<uint32_t> &slots) { check(chd_world_despawn(ctl_.ctx(), (uint32_t)slots.size(), slots.data())); }
    void SetEntityFlags(const std::vector<uint32_t> &slots, const std::vector<uint32_t> &flags) {
        check(chd_world_set_entity_flags(ctl_.ctx(), (uint32_t)slots.size(), slots.data(), flags.data()));
    }
    void RemoveSubscribers(const std::vector<uint32_t> &slots) { check(chd_subs_remove(ctl_.ctx(), (uint32_t)slots.size(), slots.data())); }
    // Connection.SubscribeToChannel with explicit ChannelSubscriptionOptions (subscription.go:34-102); shouldSend[i] as there
    Error SetSubOptions(int64_t nowNs, const std::vector<chd_sub_options> &opts, std::vector<uint8_t> &shouldSend, std::vector<int32_t> &status) {
        shouldSend.assign(opts.size(), 0);
        status.assign(opts.size(), 0);
        return err(chd_subs_set_options(ctl_.ctx(), nowNs, (uint32_t)opts.size(), opts.data(), shouldSend.data(), status.data()));
    }
    // who receives each handover's ChannelDataHandoverMessage (spatial.go:776-857): CSR over the last tick's handovers
    Error HandoverRecipients(uint32_t nHandovers, std::vector<uint32_t> &offsets, std::vector<uint32_t> &conn, std::vector<uint8_t> &kind) {
        offsets.assign((size_t)nHandovers + 1, 0);
        uint64_t n = 0;
        int rc = chd_handover_recipients(ctl_.ctx(), offsets.data(), nullptr, nullptr, 0, &n);  // count
        if (rc != CHD_OK && rc != CHD_E_CAPACITY) return err(rc);
        conn.assign(std::max<uint64_t>(n, 1), 0);
        kind.assign(std::max<uint64_t>(n, 1), 0);
        rc = chd_handover_recipients(ctl_.ctx(), offsets.data(), conn.data(), kind.data(), conn.size(), &n);
        conn.resize(n);
        kind.resize(n);
        return err(rc);
    }
    // ctl.serverConnections (spatial.go:399-424): which ConnectionId is which spatial server — what SubscribeToChannel's DataAccess in
    // the handover loop is derived from (WRITE for the entity channel's owner, spatial.go:812-817), and with it the `shouldSend` of a
    // connection whose access the handover CHANGES (subscription.go:44-57).  Empty: no connection is an owner.
    Error SetServerConnections(const std::vector<ConnectionId> &serverConnIds) {
        return err(chd_world_set_server_connections(ctl_.ctx(), (uint32_t)serverConnIds.size(), serverConnIds.empty() ? nullptr : serverConnIds.data()));
    }
    // ... and, per (recipient, entity of the handover), whether the message carries that entity's FULL state: the `shouldSend` of
    // spatial.go:797-857 (bit q of fullMask[r] = entity q of the handover's list, in message order)
    Error HandoverRecipientsEx(uint32_t nHandovers, std::vector<uint32_t> &offsets, std::vector<uint32_t> &conn, std::vector<uint8_t> &kind,
                               std::vector<uint32_t> &fullMask) {
        offsets.assign((size_t)nHandovers + 1, 0);
        uint64_t n = 0;
        int rc = chd_handover_recipients_ex(ctl_.ctx(), offsets.data(), nullptr, nullptr, nullptr, 0, &n);  // count
        if (rc != CHD_OK && rc != CHD_E_CAPACITY) return err(rc);
        conn.assign(std::max<uint64_t>(n, 1), 0);
        kind.assign(std::max<uint64_t>(n, 1), 0);
        fullMask.assign(std::max<uint64_t>(n, 1), 0);
        rc = chd_handover_recipients_ex(ctl_.ctx(), offsets.data(), conn.data(), kind.data(), fullMask.data(), conn.size(), &n);
        conn.resize(n);
        kind.resize(n);
        fullMask.resize(n);
        return err(rc);
    }
    // step 1 of a cross-server handover (spatial.go:688-694): flags[h] = the src spatial server's connection is unsubscribed from the
    // handover entities' channels (it has no interest in dst) — for the handovers of the last tick
    Error HandoverSrcOwnerUnsubscribed(uint32_t nHandovers, std::vector<uint8_t> &flags) {
        flags.assign(std::max<uint32_t>(nHandovers, 1u), 0);
        uint32_t n = 0;
        const int rc = chd_handover_src_owner_unsubscribed(ctl_.ctx(), flags.data(), (uint32_t)flags.size(), &n);
        flags.resize(n);
        return err(rc);
    }
    // The tick two deep (chd_tick_segments_begin / _end): TickSegmentsBegin enqueues tick t+1 (uploads, the tick, the segment passes;
    // no host wait); TickSegmentsEnd waits for the OLDEST tick in flight and hands out pointers into the library's page-locked block
    // of that tick (valid until the next-but-one Begin).  The arrays `in` points to (chd_host_alloc memory) must stay untouched until
    // the matching End.  At most two ticks in flight.
    Error TickSegmentsBegin(const chd_tick_in &in) { return err(chd_tick_segments_begin(ctl_.ctx(), &in)); }
    Error TickSegmentsEnd(chd_segments_block &block) { return err(chd_tick_segments_end(ctl_.ctx(), &block)); }
    // one ChannelDataHandoverMessage per distinct (handover, full-state mask): varHandover[v] / varFullMask[v] name the variants
    Error HandoverVariants(const std::vector<uint32_t> &varHandover, const std::vector<uint32_t> &varFullMask, std::vector<uint32_t> &offsets,
                           std::vector<uint8_t> &bytes, uint64_t cap) {
        offsets.assign(varHandover.size() + 1, 0);
        bytes.assign(std::max<uint64_t>(cap, 1), 0);
        uint64_t n = 0;
        const int rc = chd_handover_variants(ctl_.ctx(), (uint32_t)varHandover.size(), varHandover.data(), varFullMask.data(), offsets.data(),
                                             bytes.data(), cap, &n);
        bytes.resize(std::min<uint64_t>(n, cap));
        return err(rc);
    }
    // stage events of the next ticks: every boundary (recordKernelOnly = false), or only the pair around the record kernel, on every
    // `every`-th tick (each event idles the tick's stream for a few microseconds)
    void SetProfiling(int depth, bool recordKernelOnly = false, int every = 1) {
        check(chd_set_profiling(ctl_.ctx(), depth));
        check(chd_set_profiling_scope(ctl_.ctx(), recordKernelOnly ? (every > 1 ? CHD_PROF_RECORD_KERNEL_EVERY(every) : CHD_PROF_RECORD_KERNEL) : CHD_PROF_STAGES));
    }
    // wire-format fan-out buffers (connection.go:57-83,626-714): payloads in, per-connection packet streams out
    void WireSetPayloads(int kind, const std::vector<uint32_t> &idx, const std::vector<uint32_t> &lens, const std::vector<uint8_t> &bytes) {
        check(chd_wire_set_payloads(ctl_.ctx(), kind, (uint32_t)idx.size(), idx.data(), lens.data(), bytes.data()));
    }
    void WireSetTypeUrl(int which, const std::string &url) { check(chd_wire_set_type_url(ctl_.ctx(), which, (const uint8_t *)url.data(), (uint32_t)url.size())); }
    Error WireBuild(uint64_t &totalBytes, uint64_t &totalPackets, uint32_t &dropped) { return err(chd_wire_build(ctl_.ctx(), &totalBytes, &totalPackets, &dropped)); }
    Error WireFetch(std::vector<uint64_t> &connOff, std::vector<uint32_t> &connPackets, std::vector<uint8_t> &bytes) {
        connOff.assign((size_t)S_ + 1, 0);
        connPackets.assign(S_, 0);
        int rc = chd_wire_fetch(ctl_.ctx(), connOff.data(), connPackets.data(), nullptr, 0);  // offsets only
        if (rc != CHD_OK) return err(rc);
        bytes.assign(std::max<uint64_t>(connOff[S_], 1), 0);
        return err(chd_wire_fetch(ctl_.ctx(), connOff.data(), connPackets.data(), bytes.data(), bytes.size()));
    }
    // the two MessagePacks of every handover of the last tick (spatial.go:738-773,797-857): blob 2h / 2h+1
    Error HandoverMessages(uint32_t nHandovers, std::vector<uint32_t> &offsets, std::vector<uint8_t> &bytes, uint64_t cap) {
        offsets.assign(2 * (size_t)nHandovers + 1, 0);
        bytes.assign(std::max<uint64_t>(cap, 1), 0);
        uint64_t n = 0;
        const int rc = chd_handover_messages(ctl_.ctx(), nHandovers, offsets.data(), bytes.data(), cap, &n);
        bytes.resize(rc == CHD_OK ? n : 0);
        return err(rc);
    }
    chd_records_digest Digest(std::vector<uint64_t> *perConnection = nullptr) {
        chd_records_digest d;
        std::memset(&d, 0, sizeof d);
        if (perConnection) perConnection->assign(S_, 0);
        check(chd_tick_digest(ctl_.ctx(), &d, perConnection ? perConnection->data() : nullptr));
        return d;
    }
    void SetPipelining(bool on) { check(chd_world_set_pipelining(ctl_.ctx(), on ? 1 : 0)); }
    void Sync() { check(chd_sync(ctl_.ctx())); }

  private:
    Error err(int rc) const {
        if (rc == CHD_OK) return {};
        const char *m = chd_last_error(ctl_.ctx());
        return {rc, m ? m : ""};
    }
    void check(int rc) {
        if (rc != CHD_OK) {
            const char *m = chd_last_error(ctl_.ctx());
            throw std::runtime_error(std::string("chd error ") + std::to_string(rc) + ": " + (m ? m : ""));
        }
    }
    StaticGrid2DSpatialController &ctl_;
    uint32_t N_, S_, capq_ = 0;
    std::vector<ConnectionId> connIds_;  // ConnectionId per subscriber slot (the host registered them: AddSubscribers)
};

// ---------------------------------------------------------------------------------------------------------------------
// One rank of a region-sharded world: rank r owns the cells whose ServerIndex == r (spatial.go:336-351,
// 399-424) — the partition CreateChannels gives to spatial server r; entities live on the rank of the cell that holds them and
// migrate with their 32-byte state (the cross-server handover of spatial.go:683-700); the tick's two exchanges run inside the
// library on RCCL (chd_shard_comm_*).  The host keeps the positions (and, if owners change, the senders) of ALL entity channels in
// device arrays indexed by channel id - EntityChannelIdStart and passes the pointers; what comes out is fetched as on one GPU.
// ---------------------------------------------------------------------------------------------------------------------
class EntityGroupTable;
class ShardWorld {
  public:
    // flags: CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP lets the tick run its interest updates beside its front.
    // historyDepth + shardChannels (both or neither): ChannelData.updateMsgBuffer element for element (data.go:53-55,149-173) with the
    // update log kept by channel id on EVERY rank — then LogSpawn (every rank, every channel of the world) beside Spawn (this rank's
    // entities), SetUpdateSenders and SetUpdateArrivals (the stamps Channel.PutMessage took, channel.go:296-310).
    ShardWorld(StaticGrid2DSpatialController &ctl, uint32_t rank, uint32_t world, uint32_t maxEntities, uint32_t maxSubscribers,
               uint64_t maxRecords = 0, uint32_t flags = 0, uint32_t historyDepth = 0, uint32_t shardChannels = 0)
        : ctl_(ctl), rank_(rank), world_(world) {
        chd_world_cfg c;
        std::memset(&c, 0, sizeof c);
        c.max_entities = maxEntities;
        c.max_subscribers = maxSubscribers;
        c.max_records = maxRecords;
        c.flags = flags;
        c.history_depth = historyDepth;
        c.shard_channels = shardChannels;
        check(chd_world_create(ctl.ctx(), &c));
    }
    // rank 0 draws the id, the gateways' own control connection carries its CHD_COMM_ID_BYTES bytes, every rank joins (collective)
    static std::vector<uint8_t> CommUniqueId() {
        std::vector<uint8_t> id(CHD_COMM_ID_BYTES);
        if (chd_shard_comm_unique_id(id.data()) != CHD_OK) throw std::runtime_error("chd_shard_comm_unique_id failed");
        return id;
    }
    void CommInit(const std::vector<uint8_t> &id, uint32_t migrateCap) { check(chd_shard_comm_init(ctl_.ctx(), id.data(), rank_, world_, migrateCap)); }
    // the entities of this rank's region (entity channel ids; slots are the library's), the connections pinned to this rank
    void Spawn(const std::vector<ChannelId> &entityChannelIds, const std::vector<double> &x, const std::vector<double> &z,
               const std::vector<uint32_t> &flags, const std::vector<ConnectionId> &owner) {
        check(chd_shard_spawn(ctl_.ctx(), (uint32_t)entityChannelIds.size(), entityChannelIds.data(), x.data(), z.data(),
                              flags.empty() ? nullptr : flags.data(), owner.empty() ? nullptr : owner.data()));
    }
    // entity channels that leave the world (every rank: the same list; whichever rank holds one frees its slot)
    void Despawn(const std::vector<ChannelId> &entityChannelIds) { check(chd_shard_despawn(ctl_.ctx(), (uint32_t)entityChannelIds.size(), entityChannelIds.data())); }
    void AddSubscribers(const std::vector<ConnectionId> &conn) { check(chd_subs_add(ctl_.ctx(), (uint32_t)conn.size(), nullptr, conn.data())); }
    // who sends each channel's updates (device array by channel id; nullptr: the spawn-time owner, which migrates with the entity)
    void SetUpdateSenders(const uint32_t *dSenderByChan, uint32_t nChan) { check(chd_shard_set_update_senders(ctl_.ctx(), dSenderByChan, nChan)); }
    // worlds with an update log by channel id (shardChannels): every rank is told about every channel that comes to life, and when
    // each update was enqueued (device array of int64 ns by channel id; nullptr: the tick's own time)
    void LogSpawn(const std::vector<ChannelId> &entityChannelIds, const std::vector<double> &x, const std::vector<double> &z) {
        check(chd_shard_log_spawn(ctl_.ctx(), (uint32_t)entityChannelIds.size(), entityChannelIds.data(), x.data(), z.data()));
    }
    void SetUpdateArrivals(const int64_t *dArrivalNsByChan, uint32_t nChan) { check(chd_shard_set_update_arrivals(ctl_.ctx(), dArrivalNsByChan, nChan)); }
    Error SetHandoverLists(const EntityGroupTable &groups, uint32_t nChan);
    // one tick (collective): positions by channel id on the device, this rank's queries in dIn (device pointers)
    Error Tick(int64_t nowNs, const double *dXByChan, const double *dZByChan, const uint8_t *dHasUpdate, uint32_t nChan, const chd_tick_in &dIn) {
        const int rc = chd_shard_tick(ctl_.ctx(), nowNs, dXByChan, dZByChan, dHasUpdate, nChan, &dIn);
        if (rc == CHD_OK) return {};
        const char *m = chd_last_error(ctl_.ctx());
        return {rc, m ? m : ""};
    }
    // THIS rank's connections' share of the recipients of the whole world's handovers of the last tick (gathered from every rank in
    // rank order; spatial.go:738-857 on one gateway per server region): CSR over `handovers`, kinds as CHD_HO_*, bit 0 of fullMask =
    // the notifier goes out with its full state, srcOwnerUnsubscribed[h] = this rank holds the src server's connection and it loses
    // its subscription to the entity channel (spatial.go:688-694)
    Error HandoverRecipients(const std::vector<chd_handover_rec> &handovers, std::vector<uint32_t> &offsets, std::vector<uint32_t> &conn,
                             std::vector<uint8_t> &kind, std::vector<uint32_t> &fullMask, std::vector<uint8_t> &srcOwnerUnsubscribed) {
        const uint32_t nh = (uint32_t)handovers.size();
        offsets.assign((size_t)nh + 1, 0);
        srcOwnerUnsubscribed.assign(std::max<uint32_t>(nh, 1u), 0);
        uint64_t n = 0;
        int rc = chd_shard_handover_recipients(ctl_.ctx(), nh, handovers.data(), offsets.data(), nullptr, nullptr, nullptr, nullptr, 0, &n);  // count
        if (rc != CHD_OK && rc != CHD_E_CAPACITY) { const char *m = chd_last_error(ctl_.ctx()); return {rc, m ? m : ""}; }
        conn.assign(std::max<uint64_t>(n, 1), 0);
        kind.assign(std::max<uint64_t>(n, 1), 0);
        fullMask.assign(std::max<uint64_t>(n, 1), 0);
        rc = chd_shard_handover_recipients(ctl_.ctx(), nh, handovers.data(), offsets.data(), conn.data(), kind.data(), fullMask.data(),
                                           srcOwnerUnsubscribed.data(), conn.size(), &n);
        conn.resize(n); kind.resize(n); fullMask.resize(n); srcOwnerUnsubscribed.resize(nh);
        if (rc == CHD_OK) return {};
        const char *m = chd_last_error(ctl_.ctx());
        return {rc, m ? m : ""};
    }
    uint32_t Rank() const { return rank_; }
    uint32_t WorldSize() const { return world_; }

  private:
    void check(int rc) {
        if (rc != CHD_OK) {
            const char *m = chd_last_error(ctl_.ctx());
            throw std::runtime_error(std::string("chd error ") + std::to_string(rc) + ": " + (m ? m : ""));
        }
    }
    StaticGrid2DSpatialController &ctl_;
    uint32_t rank_, world_;
};

// ---------------------------------------------------------------------------------------------------------------------
// Entity group controllers (pkg/channeld/entity.go:58-224): one FlatEntityGroupController per ENTITY channel, each with
// a pointer to a handover group and to a lock group; group instances are SHARED between the channels a cascade has
// reached.  Notify asks the notifier's controller for GetHandoverEntities() (entity.go:197-224, spatial.go:675-679).
// These per-channel views are not equivalence classes (a locked entity does not take over the group it is added to; one
// removed from a group keeps an EMPTY group and cannot hand over until it is added again, entity_test.go:82-88), so the
// engine takes the evaluated lists: EngineLists() -> chd_world_set_handover_lists.  Same method names and argument
// meaning as the Go interface EntityGroupController (entity.go:49-56); the Python twin is channeld_amd/groups.py.
// ---------------------------------------------------------------------------------------------------------------------
enum EntityGroupType { EntityGroupType_HANDOVER = 0, EntityGroupType_LOCK = 1 };  // channeld.proto
using EntityId = uint32_t;

class EntityGroupTable {
  public:
    // an ENTITY channel was created (its controller is Initialize()d, entity.go:66-68); slot = the engine's entity slot
    void CreateChannel(EntityId id, uint32_t slot) { ctl_[id] = Ctl{-1, -1, slot}; }
    // Uninitialize (entity.go:70-78): the entity leaves its current groups, which other channels may share
    void RemoveChannel(EntityId id) {
        auto it = ctl_.find(id);
        if (it == ctl_.end()) return;
        if (it->second.handover >= 0) RemoveFromGroup(id, EntityGroupType_HANDOVER, {id});
        if (ctl_[id].lock >= 0) RemoveFromGroup(id, EntityGroupType_LOCK, {id});
        ctl_.erase(id);
    }
    // entity.go:104-158, on the controller of channel `id`
    Error AddToGroup(EntityId id, EntityGroupType t, const std::vector<EntityId> &entities) {
        if (ptr(id, t) < 0) ptr(id, t) = new_set();
        for (EntityId e : entities) {
            const int k = ptr(id, t);  // (read anew for every entity, as the Go loop reads ctl.handoverGroup)
            sets_[k].insert({e, true});
            if (ctl_.count(e)) cascade(e, t, k);  // GetChannel(e) != nil: the member's controller joins the shared instance
        }
        return {};
    }
    // entity.go:160-195; the reference's error when the group pointer is nil
    Error RemoveFromGroup(EntityId id, EntityGroupType t, const std::vector<EntityId> &entities) {
        if (ptr(id, t) < 0)
            return {CHD_E_STATE, std::string(t == EntityGroupType_HANDOVER ? "handover" : "lock") + " group is nil, entityId: " + std::to_string(id)};
        for (EntityId e : entities) {
            sets_[ptr(id, t)].erase(e);  // (once the channel removes ITSELF its pointer is a fresh empty group)
            if (ctl_.count(e)) ptr(e, t) = new_set();  // the removed entity's channel starts over with an EMPTY group
        }
        return {};
    }
    // entity.go:197-224 (entity ids, ascending)
    std::vector<EntityId> GetHandoverEntities(EntityId id) const {
        const Ctl &c = ctl_.at(id);
        if (c.handover < 0) return {id};  // "If AddToGroup is never called, return the entity itself"
        const auto &members = sets_[c.handover];
        if (c.lock >= 0)
            for (const auto &kv : members)
                if (sets_[c.lock].count(kv.first)) return {};  // a locked member: the handover should not happen
        std::vector<EntityId> out;
        for (const auto &kv : members) out.push_back(kv.first);
        return out;
    }
    // what chd_world_set_handover_lists takes: every channel's evaluated list, members as entity slots (ids without an
    // entity channel have no engine state and are left out), identical lists shared
    struct Lists {
        std::vector<uint32_t> list_off{0}, list_members, idx, list_of;
    };
    Lists EngineLists() const {
        Lists L;
        std::map<std::vector<uint32_t>, uint32_t> seen;
        std::vector<std::vector<uint32_t>> order;
        for (const auto &kv : ctl_) {  // ascending entity id
            L.idx.push_back(kv.second.slot);
            if (kv.second.handover < 0) { L.list_of.push_back(CHD_NO_HANDOVER_LIST); continue; }
            std::vector<uint32_t> key;
            for (EntityId m : GetHandoverEntities(kv.first)) {
                auto it = ctl_.find(m);
                if (it != ctl_.end()) key.push_back(it->second.slot);
            }
            std::sort(key.begin(), key.end());
            auto ins = seen.insert({key, (uint32_t)order.size()});
            if (ins.second) order.push_back(key);
            L.list_of.push_back(ins.first->second);
        }
        for (const auto &key : order) {
            L.list_members.insert(L.list_members.end(), key.begin(), key.end());
            L.list_off.push_back((uint32_t)L.list_members.size());
        }
        return L;
    }
    // the same lists keyed by ENTITY CHANNEL ID (an entity's id is its channel id): what a region-sharded world takes
    // (chd_shard_set_handover_lists — slots are the library's there); Lists::idx then holds channel ids
    Lists ShardLists() const {
        Lists L;
        std::map<std::vector<uint32_t>, uint32_t> seen;
        std::vector<std::vector<uint32_t>> order;
        for (const auto &kv : ctl_) {
            L.idx.push_back(kv.first);
            if (kv.second.handover < 0) { L.list_of.push_back(CHD_NO_HANDOVER_LIST); continue; }
            std::vector<uint32_t> key;
            for (EntityId m : GetHandoverEntities(kv.first))
                if (ctl_.count(m)) key.push_back(m);
            std::sort(key.begin(), key.end());
            auto ins = seen.insert({key, (uint32_t)order.size()});
            if (ins.second) order.push_back(key);
            L.list_of.push_back(ins.first->second);
        }
        for (const auto &key : order) {
            L.list_members.insert(L.list_members.end(), key.begin(), key.end());
            L.list_off.push_back((uint32_t)L.list_members.size());
        }
        return L;
    }
    Error UploadShard(chd_ctx *ctx, uint32_t nChan) const {  // (every rank: the same, whole-world lists)
        const Lists L = ShardLists();
        const uint32_t n_lists = (uint32_t)L.list_off.size() - 1;
        const int rc = chd_shard_set_handover_lists(ctx, n_lists, n_lists ? L.list_off.data() : nullptr,
                                                    L.list_members.empty() ? nullptr : L.list_members.data(), (uint32_t)L.idx.size(),
                                                    L.idx.empty() ? nullptr : L.idx.data(), L.list_of.empty() ? nullptr : L.list_of.data(), nChan);
        if (rc == CHD_OK) return {};
        const char *m = chd_last_error(ctx);
        return {rc, m ? m : ""};
    }
    Error Upload(chd_ctx *ctx) const {
        const Lists L = EngineLists();
        const uint32_t n_lists = (uint32_t)L.list_off.size() - 1;
        const int rc = chd_world_set_handover_lists(ctx, n_lists, n_lists ? L.list_off.data() : nullptr,
                                                    L.list_members.empty() ? nullptr : L.list_members.data(), (uint32_t)L.idx.size(),
                                                    L.idx.empty() ? nullptr : L.idx.data(), L.list_of.empty() ? nullptr : L.list_of.data());
        if (rc == CHD_OK) return {};
        const char *m = chd_last_error(ctx);
        return {rc, m ? m : ""};
    }

  private:
    struct Ctl { int handover, lock; uint32_t slot; };
    int &ptr(EntityId id, EntityGroupType t) { return t == EntityGroupType_HANDOVER ? ctl_.at(id).handover : ctl_.at(id).lock; }
    int new_set() { sets_.emplace_back(); return (int)sets_.size() - 1; }
    void add_all(int dst, int src) {
        if (src < 0 || src == dst) return;
        for (const auto &kv : sets_[src]) sets_[dst].insert(kv);
    }
    // cascadeGroup (entity.go:80-102) on channel e's controller with the shared instance k
    void cascade(EntityId e, EntityGroupType t, int k) {
        Ctl &c = ctl_.at(e);
        if (c.lock >= 0 && !sets_[c.lock].empty()) return;  // "Current entity is already locked, won't cascade."
        if (t == EntityGroupType_HANDOVER) {
            add_all(k, c.handover);
            c.handover = k;
        } else {  // LOCK outranks HANDOVER: the cascade brings the handover group's entities into the lock group
            add_all(k, c.handover);
            add_all(k, c.lock);
            c.lock = k;
        }
    }
    std::map<EntityId, Ctl> ctl_;
    std::vector<std::map<EntityId, bool>> sets_;  // arena of group instances: "two channels share one group" = same index
};

inline Error ShardWorld::SetHandoverLists(const EntityGroupTable &groups, uint32_t nChan) { return groups.UploadShard(ctl_.ctx(), nChan); }

}  // namespace chd

// Exercises include/chd_spatial.hpp (the C++ host mirror of the Go SpatialController interface) — driven by
// tests/test_cxx_host.py, which compares every line with what the Python mirror gets through the same C-ABI.
//   host_mirror_check cos <x>...            go_cos bit patterns
//   host_mirror_check pack                  the packed chd_aoi_query records of a fixed query set
//   host_mirror_check batch <exact> <events>  UpdateBatch::Layout of a script: u,slot,x,z,sender,arrival | c,channel,sender,arrival | q,sub,index
//   host_mirror_check groups < script     EntityGroupTable driven by a script on stdin
//   host_mirror_check load <config.json>    LoadConfig only: prints the error code (CHD_E_NO_DEVICE without a GPU)
//   host_mirror_check gpu <config.json>     the interface methods on golden inputs + one world of three ticks
#include <cinttypes>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "chd_spatial.hpp"

using namespace chd;

static std::vector<SpatialInterestQuery> fixed_queries() {
    std::vector<SpatialInterestQuery> qs(5);
    qs[0].SphereAOI = SphereAOI{SpatialInfo{10.5, 0, -20.25}, 150.0};
    qs[1].BoxAOI = BoxAOI{SpatialInfo{4.9, 0, 4.9}, SpatialInfo{4.9, 0, 10.0}};
    qs[2].ConeAOI = ConeAOI{SpatialInfo{-1500.0, 0, 250.0}, SpatialInfo{0.6, 0, -0.8}, 0.5236, 30000.0};
    qs[3].SpotsAOI = SpotsAOI{{SpatialInfo{1, 0, 2}, SpatialInfo{-3, 0, 4.5}, SpatialInfo{1e6, 0, 0}}, {7, 0}};
    qs[4].SpotsAOI = SpotsAOI{{SpatialInfo{100, 0, 100}}, {}};
    qs[4].SphereAOI = SphereAOI{SpatialInfo{0, 0, 0}, 3000.0};
    qs[4].ConeAOI = ConeAOI{SpatialInfo{0, 0, 0}, SpatialInfo{1, 0, 0}, 0.1, 6000.0};
    return qs;
}

static void hex(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) std::printf("%02x", b[i]);
    std::printf("\n");
}

static std::string slurp(const char *path) {
    std::ifstream f(path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "cos") {
        for (int i = 2; i < argc; i++) {
            const double v = go_cos(std::strtod(argv[i], nullptr));
            uint64_t bits;
            std::memcpy(&bits, &v, 8);
            std::printf("%016" PRIx64 "\n", bits);
        }
        return 0;
    }
    if (mode == "batch") {
        // the same script goes through channeld_amd.engine.UpdateBatch in tests/test_cxx_host.py
        UpdateBatch b(std::atoi(argv[2]) != 0);
        auto qs = fixed_queries();
        for (int i = 3; i < argc; i++) {
            std::vector<std::string> f;
            std::stringstream ss(argv[i]);
            for (std::string t; std::getline(ss, t, ',');) f.push_back(t);
            if (f[0] == "u") b.OnUpdate((uint32_t)std::stoul(f[1]), std::stod(f[2]), std::stod(f[3]), (uint32_t)std::stoul(f[4]), std::stoll(f[5]));
            else if (f[0] == "c") b.OnCellUpdate((uint32_t)std::stoul(f[1]), (uint32_t)std::stoul(f[2]), std::stoll(f[3]));
            else if (f[0] == "q") b.OnInterest((uint32_t)std::stoul(f[1]), &qs[std::stoul(f[2]) % qs.size()]);
        }
        b.Layout();
        auto line = [](const char *name, auto &v) {
            std::printf("%s", name);
            for (auto e : v) std::printf(" %lld", (long long)e);
            std::printf("\n");
        };
        line("slot", b.updSlot); line("sender", b.updSender); line("round_off", b.roundOff); line("arrival", b.updArrivalNs);
        hex(b.updX.data(), b.updX.size() * 8); hex(b.updZ.data(), b.updZ.size() * 8);
        line("cell", b.cellChannel); line("cell_sender", b.cellSender); line("cell_arrival", b.cellArrivalNs);
        line("query_sub", b.querySub);
        std::printf("queries");
        for (auto *q : b.queries) std::printf(" %lld", (long long)(q - qs.data()));
        std::printf("\n");
        return 0;
    }
    if (mode == "pack") {
        auto qs = fixed_queries();
        std::vector<const SpatialInterestQuery *> ptrs;
        for (auto &q : qs) ptrs.push_back(&q);
        PackedQueries p;
        if (Error e = pack_queries(ptrs, p)) { std::printf("error %d\n", e.code); return 1; }
        hex(p.q.data(), p.q.size() * sizeof(chd_aoi_query));
        hex(p.spot_x.data(), 8 * p.spot_x.size());
        hex(p.spot_z.data(), 8 * p.spot_z.size());
        hex(p.spot_dist.data(), 4 * p.spot_dist.size());
        SpatialInterestQuery bad;
        bad.BoxAOI = BoxAOI{SpatialInfo{0, 0, 0}, std::nullopt};
        std::printf("nil-extent %d\n", pack_queries({&bad}, p).code);
        std::printf("nil-query %d\n", pack_queries({nullptr}, p).code);
        return 0;
    }
    if (mode == "groups") {
        // script on stdin: "C id slot" create channel, "A id type n e..." AddToGroup, "R id type n e..." RemoveFromGroup,
        // "D id" remove channel, "G id" print GetHandoverEntities, "L" print the engine lists
        EntityGroupTable t;
        char op;
        while (std::scanf(" %c", &op) == 1) {
            if (op == 'C') { unsigned id, slot; if (std::scanf("%u %u", &id, &slot) != 2) return 3; t.CreateChannel(id, slot); }
            else if (op == 'D') { unsigned id; if (std::scanf("%u", &id) != 1) return 3; t.RemoveChannel(id); }
            else if (op == 'A' || op == 'R') {
                unsigned id, ty, n;
                if (std::scanf("%u %u %u", &id, &ty, &n) != 3) return 3;
                std::vector<EntityId> es(n);
                for (auto &e : es) if (std::scanf("%u", &e) != 1) return 3;
                Error err = op == 'A' ? t.AddToGroup(id, (EntityGroupType)ty, es) : t.RemoveFromGroup(id, (EntityGroupType)ty, es);
                std::printf("%c %d\n", op, err.code);
            } else if (op == 'G') {
                unsigned id;
                if (std::scanf("%u", &id) != 1) return 3;
                std::printf("G %u:", id);
                for (EntityId e : t.GetHandoverEntities(id)) std::printf(" %u", e);
                std::printf("\n");
            } else if (op == 'L') {
                auto L = t.EngineLists();
                auto dump = [](const char *n, const std::vector<uint32_t> &v) { std::printf("%s", n); for (uint32_t x : v) std::printf(" %u", x); std::printf("\n"); };
                dump("off", L.list_off); dump("mem", L.list_members); dump("idx", L.idx); dump("of", L.list_of);
            } else return 3;
        }
        return 0;
    }
    if (argc < 3) return 2;
    const std::string cfg = slurp(argv[2]);
    StaticGrid2DSpatialController ctl;
    if (mode == "load") {
        Error e = ctl.LoadConfig(cfg, false);
        std::printf("load %d\n", e.code);
        Error bad = ctl.LoadConfig("{\"GridWidth\": \"wide\"}", false);
        std::printf("badjson %d\n", bad.code);
        Error neg = ctl.LoadConfig("{\"GridWidth\": 10, \"GridHeight\": 10, \"GridCols\": -2, \"GridRows\": 1, \"ServerCols\": 1, \"ServerRows\": 1}", false);
        std::printf("negcols %d\n", neg.code);
        return 0;
    }
    if (mode != "gpu") return 2;
    if (Error e = ctl.LoadConfig(cfg, false)) { std::printf("load %d %s\n", e.code, e.msg.c_str()); return 1; }
    std::printf("grid %u %u %u %u %u %.17g %.17g %.17g %.17g\n", ctl.GridCols, ctl.GridRows, ctl.ServerCols, ctl.ServerRows,
                ctl.ServerInterestBorderSize, ctl.GridWidth, ctl.GridHeight, ctl.WorldOffsetX, ctl.WorldOffsetZ);
    // GetChannelId on a lattice of points incl. the edges and one point outside
    const double W = ctl.GridWidth * ctl.GridCols, H = ctl.GridHeight * ctl.GridRows;
    for (int i = 0; i <= 8; i++) {
        SpatialInfo p{ctl.WorldOffsetX + W * i / 8.0, 0, ctl.WorldOffsetZ + H * (8 - i) / 8.0 - (i == 0 ? 1e-9 * H : 0)};
        auto r = ctl.GetChannelId(p);
        std::printf("id %u %d\n", r.first, r.second.code);
    }
    // QueryChannelIds: a sphere, a box, a cone around the world centre; an invalid radius; a centre outside
    const double cx = ctl.WorldOffsetX + W / 2, cz = ctl.WorldOffsetZ + H / 2;
    std::vector<SpatialInterestQuery> qs(5);
    qs[0].SphereAOI = SphereAOI{SpatialInfo{cx, 0, cz}, 1.5 * ctl.GridWidth};
    qs[1].BoxAOI = BoxAOI{SpatialInfo{cx, 0, cz}, SpatialInfo{ctl.GridWidth, 0, 2 * ctl.GridHeight}};
    qs[2].ConeAOI = ConeAOI{SpatialInfo{cx, 0, cz}, SpatialInfo{0.6, 0, 0.8}, 0.5236, 3 * ctl.GridWidth};
    qs[3].SphereAOI = SphereAOI{SpatialInfo{cx, 0, cz}, -1.0};
    qs[4].SphereAOI = SphereAOI{SpatialInfo{ctl.WorldOffsetX - 10, 0, cz}, ctl.GridWidth};
    for (auto &q : qs) {
        auto r = ctl.QueryChannelIds(&q);
        std::printf("aoi %d", r.second.code);
        for (auto &kv : r.first) std::printf(" %u:%u", kv.first, kv.second);
        std::printf("\n");
    }
    auto regions = ctl.GetRegions();
    std::printf("regions %zu", regions.first.size());
    for (size_t i = 0; i < regions.first.size(); i += std::max<size_t>(1, regions.first.size() / 5))
        std::printf(" %u:%u:%.17g:%.17g", regions.first[i].ChannelId_, regions.first[i].ServerIndex, regions.first[i].Min.X, regions.first[i].Max.Z);
    std::printf("\n");
    for (ChannelId c : {SpatialChannelIdStart, SpatialChannelIdStart + ctl.GridCols * ctl.GridRows / 2, SpatialChannelIdStart + ctl.GridCols * ctl.GridRows - 1}) {
        auto adj = ctl.GetAdjacentChannels(c);
        std::printf("adj %u:", c);
        for (ChannelId a : adj.first) std::printf(" %u", a);
        std::printf("\n");
    }
    auto own = ctl.CreateChannels(501);
    std::printf("server0 %zu first %u last %u next %u\n", own.first.size(), own.first.empty() ? 0 : own.first.front(),
                own.first.empty() ? 0 : own.first.back(), ctl.nextServerIndex());
    int calls = 0;
    ctl.Notify(SpatialInfo{cx - 1, 0, cz - 1}, SpatialInfo{cx + ctl.GridWidth, 0, cz - 1}, [&](ChannelId s, ChannelId d, void *) { calls++; std::printf("notify %u %u\n", s, d); });
    ctl.Notify(SpatialInfo{cx - 1, 0, cz - 1}, SpatialInfo{cx - 2, 0, cz - 1}, [&](ChannelId, ChannelId, void *) { calls++; });
    std::printf("notify-calls %d\n", calls);
    // a small world: 64 entities on a diagonal, 4 connections with sphere interests, three ticks 50 ms apart
    const uint32_t N = 64, S = 4;
    SpatialWorld world(ctl, N, S, 1u << 16);
    std::vector<ChannelId> ids(N);
    std::vector<double> x(N), z(N);
    std::vector<uint32_t> fl(N, 0), owner(N, 1);
    for (uint32_t i = 0; i < N; i++) {
        ids[i] = EntityChannelIdStart + i;
        x[i] = ctl.WorldOffsetX + W * (i + 0.5) / N;
        z[i] = ctl.WorldOffsetZ + H * (i + 0.5) / N;
    }
    world.Spawn(ids, x, z, fl, owner);
    world.AddSubscribers({1000, 1001, 1002, 1003});
    for (int t = 1; t <= 3; t++) {
        for (uint32_t i = 0; i < N; i++) x[i] += 0.3 * ctl.GridWidth;  // everybody walks east: some cross a cell border
        for (uint32_t i = 0; i < N; i++) if (x[i] >= ctl.WorldOffsetX + W) x[i] -= W;
        std::vector<SpatialInterestQuery> q(S);
        std::vector<const SpatialInterestQuery *> qp;
        for (uint32_t s = 0; s < S; s++) {
            q[s].SphereAOI = SphereAOI{SpatialInfo{x[s * 16], 0, z[s * 16]}, 1.2 * ctl.GridWidth};
            qp.push_back(&q[s]);
        }
        SpatialWorld::TickResult r;
        Error e = world.Tick((int64_t)t * 50000000, x, z, qp, 1u << 16, r);
        std::printf("tick %d rc %d handovers %zu aborts %u unsubs %zu newsubs %zu records %zu overflow %u\n", t, e.code, r.handovers.size(),
                    r.lockedAborts, r.unsubSlot.size(), r.newSubSlot.size(), r.records.size(), r.overflow);
        uint64_t hsum = 0;
        for (auto &h : r.handovers) hsum += (uint64_t)h.entity * 1315423911u + h.src * 31u + h.dst;
        uint64_t rsum = 0;
        for (auto &rec : r.records) rsum += ((uint64_t)rec.conn << 32 | rec.channel) * 0x9E3779B97F4A7C15ull;
        std::printf("digest %" PRIu64 " %" PRIu64, hsum, rsum);
        for (uint32_t s = 0; s < S; s++) std::printf(" %u", r.connRecordCount[s]);
        std::printf("\n");
    }
    {
        // a fourth tick from single messages (UpdateBatch, ring world): every third entity sends two updates — the last one counts —,
        // connection 1 re-queries twice (the later query counts), connection 3 once, the others not at all
        UpdateBatch b(false);
        for (uint32_t i = 0; i < N; i += 3) b.OnUpdate(i, x[i] + 0.1 * ctl.GridWidth, z[i], 1, 180000000);
        for (uint32_t i = 0; i < N; i += 3) {
            x[i] += 0.6 * ctl.GridWidth;
            if (x[i] >= ctl.WorldOffsetX + W) x[i] -= W;
            b.OnUpdate(i, x[i], z[i], 1, 190000000);
        }
        std::vector<SpatialInterestQuery> q(3);
        q[0].SphereAOI = SphereAOI{SpatialInfo{x[16], 0, z[16]}, 1.2 * ctl.GridWidth};
        q[1].SphereAOI = SphereAOI{SpatialInfo{x[48], 0, z[48]}, 2.0 * ctl.GridWidth};
        q[2].SphereAOI = SphereAOI{SpatialInfo{x[16], 0, z[16]}, 0.8 * ctl.GridWidth};
        b.OnInterest(1, &q[0]);
        b.OnInterest(3, &q[1]);
        b.OnInterest(1, &q[2]);
        SpatialWorld::TickResult r;
        Error e = world.Tick((int64_t)200000000, b, 1u << 16, r);
        std::printf("tick 4 rc %d handovers %zu aborts %u unsubs %zu newsubs %zu records %zu overflow %u\n", e.code, r.handovers.size(),
                    r.lockedAborts, r.unsubSlot.size(), r.newSubSlot.size(), r.records.size(), r.overflow);
        uint64_t hsum = 0;
        for (auto &h : r.handovers) hsum += (uint64_t)h.entity * 1315423911u + h.src * 31u + h.dst;
        uint64_t rsum = 0;
        for (auto &rec : r.records) rsum += ((uint64_t)rec.conn << 32 | rec.channel) * 0x9E3779B97F4A7C15ull;
        std::printf("digest %" PRIu64 " %" PRIu64, hsum, rsum);
        for (uint32_t s = 0; s < S; s++) std::printf(" %u", r.connRecordCount[s]);
        std::printf("\n");
    }
    return 0;
}

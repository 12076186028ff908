// Host-runtime harness for the sanitizer builds (tools/sanitize_host.py): exercises the parts
// of the native runtime that carry state machines and run without a GPU - the TCP rendezvous
// store (server thread + concurrent clients, malformed and oversize requests, time-outs), the
// bucket planner and the engine's layout planner - under AddressSanitizer + UBSan, and the
// store again under ThreadSanitizer.  Exit code 0 = every check held.
//
//   host_harness store | plan | engine | all
#include <arpa/inet.h>
#include <netinet/in.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dml_comm.h"
#ifndef HARNESS_STORE_ONLY
#include "dml_engine.h"
#endif

using namespace dml;

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

static void raw_request(int port, const std::string& bytes) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    struct sockaddr_in a;
    memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    ::inet_pton(AF_INET, "127.0.0.1", &a.sin_addr);
    if (::connect(fd, (struct sockaddr*)&a, sizeof(a)) == 0) {
        (void)::send(fd, bytes.data(), bytes.size(), MSG_NOSIGNAL);
        char buf[16];
        struct timeval tv = {1, 0};
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        (void)::recv(fd, buf, sizeof(buf), 0);
    }
    ::close(fd);
}

static void test_store() {
    const int world = 4;
    TcpStore server("127.0.0.1", 0, true, world, 20.0);
    const int port = server.port();
    std::vector<std::string> got(world);
    std::vector<std::thread> ts;
    for (int r = 1; r < world; ++r)
        ts.emplace_back([&, r] {
            TcpStore c("127.0.0.1", port, false, world, 20.0);
            got[r] = c.get("uid");                  // parks until rank 0 publishes
            for (int i = 0; i < 50; ++i) c.add("ctr", 1);
            c.set("k" + std::to_string(r), std::string(1000 * r, (char)('a' + r)));
            c.barrier("done");
        });
    // hostile traffic while the ranks are parked: a 3 GiB key announcement, a truncated
    // request, an unknown op
    std::string big(1, (char)1);
    uint32_t huge = 3u << 30;
    big.append((const char*)&huge, 4);
    raw_request(port, big);
    raw_request(port, std::string("\x02\x05\x00", 3));
    std::string unk(1, (char)9);
    uint32_t z = 0;
    unk.append((const char*)&z, 4);
    unk.append((const char*)&z, 4);
    raw_request(port, unk);
    std::string uid(128, '\0');
    for (int i = 0; i < 128; ++i) uid[i] = (char)(i * 7);
    server.set("uid", uid);
    server.barrier("done");
    for (auto& t : ts) t.join();
    for (int r = 1; r < world; ++r) CHECK(got[r] == uid);
    CHECK(server.add("ctr", 0) == 50 * (world - 1));
    CHECK(server.get("k2") == std::string(2000, 'c'));
    // time-outs are errors, not hangs
    TcpStore lonely("127.0.0.1", 0, true, 2, 0.3);
    bool threw = false;
    try { lonely.get("never"); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { lonely.barrier("alone"); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    std::printf("store ok\n");
}

#ifndef HARNESS_STORE_ONLY
static void test_plan() {
    const size_t el[] = {1728, 64, 73728, 128, 294912, 256, 589824, 256, 1179648, 512,
                         2359296, 512, 2359296, 512, 2359296, 512, 5120, 10};
    std::vector<size_t> sizes;
    for (size_t e : el) sizes.push_back(4 * e);
    for (size_t cap_mb : {1, 2, 5, 10, 25, 100}) {
        for (size_t first_mb : {0, 1, 10}) {
            auto b = plan_buckets(sizes, cap_mb << 20, first_mb << 20);
            CHECK(!b.empty());
            int hi = (int)sizes.size();      // completion order: back to front, no gaps
            for (auto& r : b) {
                CHECK(r.hi == hi && r.lo < r.hi && r.lo >= 0);
                hi = r.lo;
            }
            CHECK(hi == 0);
        }
    }
    CHECK(plan_buckets({}, 1 << 20, 0).empty());
    std::vector<ManualSync::Slice> t = {{0, 5}, {64, 7}, {128, 1}};
    CHECK(ManualSync::packed_elems(t) == 16);
    std::printf("plan ok\n");
}

static void test_engine() {
    const std::vector<std::vector<int>> tables = {
        {64, -1, 128, -1, 256, 256, -1, 512, 512, -1, 512, 512, -1},                       // VGG11
        {64, 64, -1, 128, 128, -1, 256, 256, -1, 512, 512, -1, 512, 512, -1},              // VGG13
        {64, 64, -1, 128, 128, -1, 256, 256, 256, 256, -1, 512, 512, 512, 512, -1, 512, 512, 512,
         512, -1}};                                                                         // VGG19
    for (auto& w : tables)
        for (bool bn : {false, true})
            for (int batch : {1, 8, 64, 256}) {
                VggEngine e(w, batch, bn, 10);
                size_t end = 0;
                for (auto& s : e.param_slots()) {
                    CHECK(s.offset % 64 == 0 && s.offset >= end);
                    end = s.offset + s.numel;
                }
                CHECK(end <= e.param_elems() && e.param_elems() % 4 == 0);
                CHECK(e.arena_elems() > e.param_elems() && e.workspace_elems() > 0);
                CHECK((e.bn_elems() > 0) == bn);
                auto table = e.layer_table();
                CHECK(table.size() == (size_t)e.num_conv());
                for (auto& row : table)
                    for (int k = 4; k < 7; ++k) CHECK((size_t)row[k] <= e.arena_elems());
            }
    for (auto bad : std::vector<std::vector<int>>{{}, {-1, 64}, {64, 100}, {128}, {64, -1, -1}}) {
        bool threw = false;
        try { VggEngine e(bad, 8, false, 10); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
    }
    std::printf("engine ok\n");
}
#endif

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "store" || what == "all") test_store();
#ifndef HARNESS_STORE_ONLY
    if (what == "plan" || what == "all") test_plan();
    if (what == "engine" || what == "all") test_engine();
#endif
    return 0;
}

// TCP key-value store used to publish the RCCL unique id at --master-ip host:port.
//
// The reference rendezvous is torch's TCPStore behind
// init_process_group(init_method="tcp://" + master_ip) (/root/reference/part2/2a/main.py:197);
// this is the native equivalent with the three operations the bootstrap needs.
//
// Wire format (little endian): request = op:u8, klen:u32, key, vlen:u32, value;
// reply = status:u8 (0 ok, 1 timeout), vlen:u32, value.  Ops: 1 SET, 2 GET (blocks on the
// server until the key exists), 3 ADD (value = i64 delta, reply = i64 new value).
// The server is a single poll() loop; blocked GETs are parked and answered by the SET
// that creates their key.  It is a rendezvous helper, not a database: keys and values are
// capped at 64 KiB (a connection that announces more is dropped), replies are bounded by a
// send time-out so one stalled client cannot freeze the loop, and the listener binds the
// --master-ip host when that is an address of this machine.
#include <arpa/inet.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <stdexcept>

#include "dml_comm.h"

namespace dml {

namespace {

using Clock = std::chrono::steady_clock;

double seconds_since(Clock::time_point t0) {
    return std::chrono::duration<double>(Clock::now() - t0).count();
}

bool send_all(int fd, const void* buf, size_t n) {
    const char* p = (const char*)buf;
    while (n) {
        ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}

// returns false on EOF / error / deadline
bool recv_all(int fd, void* buf, size_t n, double timeout_s) {
    char* p = (char*)buf;
    const auto t0 = Clock::now();
    while (n) {
        struct pollfd pf = {fd, POLLIN, 0};
        const double left = timeout_s - seconds_since(t0);
        if (left <= 0) return false;
        int pr = ::poll(&pf, 1, (int)(left * 1000) + 1);
        if (pr < 0 && errno == EINTR) continue;
        if (pr <= 0) return false;
        ssize_t k = ::recv(fd, p, n, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        p += k;
        n -= (size_t)k;
    }
    return true;
}

constexpr uint32_t kMaxField = 64u << 10;      // key / value size cap, bytes

struct Conn {
    int fd;
    std::string inbuf;
    bool waiting = false;       // parked GET
    std::string wait_key;
};

void reply(int fd, uint8_t status, const std::string& val) {
    std::string out;
    out.push_back((char)status);
    uint32_t n = (uint32_t)val.size();
    out.append((const char*)&n, 4);
    out.append(val);
    send_all(fd, out.data(), out.size());
}

}  // namespace

TcpStore::TcpStore(const std::string& host, int port, bool is_server, int world_size,
                   double timeout_s)
    : host_(host), port_(port), is_server_(is_server), world_size_(world_size),
      timeout_s_(timeout_s) {
    bool bound_to_host = false;
    if (is_server_) {
        listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listen_fd_ < 0) throw std::runtime_error("TcpStore: socket() failed");
        int one = 1;
        ::setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        struct sockaddr_in addr;
        memset(&addr, 0, sizeof(addr));
        addr.sin_family = AF_INET;
        addr.sin_port = htons((uint16_t)port_);
        // listen on the rendezvous address itself when it is a local IPv4 address (the
        // reference's default 127.0.1.1, or this node's address), on every interface otherwise
        bool bound = false;
        if (::inet_pton(AF_INET, host_.c_str(), &addr.sin_addr) == 1)
            bound = ::bind(listen_fd_, (struct sockaddr*)&addr, sizeof(addr)) == 0;
        bound_to_host = bound;
        if (!bound) addr.sin_addr.s_addr = htonl(INADDR_ANY);
        if (!bound && ::bind(listen_fd_, (struct sockaddr*)&addr, sizeof(addr)) != 0) {
            ::close(listen_fd_);
            throw std::runtime_error("TcpStore: cannot bind port " + std::to_string(port_) + ": " +
                                     strerror(errno));
        }
        if (port_ == 0) {
            socklen_t len = sizeof(addr);
            ::getsockname(listen_fd_, (struct sockaddr*)&addr, &len);
            port_ = ntohs(addr.sin_port);
        }
        ::listen(listen_fd_, 128);
        server_ = std::thread([this] { serve(); });
    }
    // client side (the server process connects to itself over loopback)
    struct addrinfo hints, *res = nullptr;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string target = (is_server_ && !bound_to_host) ? std::string("127.0.0.1") : host_;
    const std::string port_s = std::to_string(port_);
    if (::getaddrinfo(target.c_str(), port_s.c_str(), &hints, &res) != 0 || !res) {
        stop_ = true;
        if (server_.joinable()) server_.join();
        throw std::runtime_error("TcpStore: cannot resolve " + target);
    }
    const auto t0 = Clock::now();
    while (true) {
        client_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (client_fd_ >= 0 && ::connect(client_fd_, res->ai_addr, res->ai_addrlen) == 0) break;
        if (client_fd_ >= 0) ::close(client_fd_);
        client_fd_ = -1;
        if (seconds_since(t0) > timeout_s_) {
            ::freeaddrinfo(res);
            stop_ = true;
            if (server_.joinable()) server_.join();
            throw std::runtime_error("TcpStore: timed out after " + std::to_string(timeout_s_) +
                                     " s connecting to " + target + ":" + port_s);
        }
        ::usleep(50 * 1000);
    }
    ::freeaddrinfo(res);
    int one = 1;
    ::setsockopt(client_fd_, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}

TcpStore::~TcpStore() {
    if (client_fd_ >= 0) {
        ::shutdown(client_fd_, SHUT_RDWR);
        ::close(client_fd_);
        client_fd_ = -1;
    }
    if (is_server_) {
        // let the other ranks finish and disconnect before the listener goes away
        const auto t0 = Clock::now();
        while (live_clients_.load() > 0 && seconds_since(t0) < 10.0) ::usleep(10 * 1000);
        stop_ = true;
        if (server_.joinable()) server_.join();
        if (listen_fd_ >= 0) ::close(listen_fd_);
    }
}

void TcpStore::serve() {
    std::map<std::string, std::string> kv;
    std::vector<Conn> conns;
    while (!stop_.load()) {
        std::vector<struct pollfd> pfs;
        pfs.push_back({listen_fd_, POLLIN, 0});
        for (auto& c : conns) pfs.push_back({c.fd, POLLIN, 0});
        const int pr = ::poll(pfs.data(), pfs.size(), 50);
        if (pr <= 0) continue;
        if (pfs[0].revents & POLLIN) {
            int fd = ::accept(listen_fd_, nullptr, nullptr);
            if (fd >= 0) {
                int one = 1;
                ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                struct timeval tv = {2, 0};     // a reply never blocks the loop for longer
                ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
                Conn c;
                c.fd = fd;
                conns.push_back(c);
                live_clients_.fetch_add(1);
            }
        }
        for (size_t i = 1; i < pfs.size(); ++i) {
            if (!(pfs[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
            Conn& c = conns[i - 1];
            char tmp[4096];
            ssize_t k = ::recv(c.fd, tmp, sizeof(tmp), 0);
            if (k <= 0) {
                ::close(c.fd);
                c.fd = -1;
                live_clients_.fetch_sub(1);
                continue;
            }
            c.inbuf.append(tmp, (size_t)k);
            // parse as many complete requests as are buffered
            while (true) {
                if (c.inbuf.size() < 5) break;
                const uint8_t op = (uint8_t)c.inbuf[0];
                uint32_t klen;
                memcpy(&klen, c.inbuf.data() + 1, 4);
                bool oversize = klen > kMaxField;
                uint32_t vlen = 0;
                if (!oversize) {
                    if (c.inbuf.size() < 5 + (size_t)klen + 4) break;
                    memcpy(&vlen, c.inbuf.data() + 5 + klen, 4);
                    oversize = vlen > kMaxField;
                }
                if (oversize) {      // not one of ours: drop the connection, keep serving
                    reply(c.fd, 1, "");
                    ::close(c.fd);
                    c.fd = -1;
                    live_clients_.fetch_sub(1);
                    break;
                }
                const size_t need = 5 + (size_t)klen + 4 + vlen;
                if (c.inbuf.size() < need) break;
                const std::string key = c.inbuf.substr(5, klen);
                const std::string val = c.inbuf.substr(5 + klen + 4, vlen);
                c.inbuf.erase(0, need);
                if (op == 1) {
                    kv[key] = val;
                    reply(c.fd, 0, "");
                    for (auto& w : conns) {
                        if (w.fd >= 0 && w.waiting && w.wait_key == key) {
                            w.waiting = false;
                            reply(w.fd, 0, val);
                        }
                    }
                } else if (op == 2) {
                    auto it = kv.find(key);
                    if (it != kv.end()) {
                        reply(c.fd, 0, it->second);
                    } else {
                        c.waiting = true;
                        c.wait_key = key;
                    }
                } else if (op == 3) {
                    int64_t delta = 0, cur = 0;
                    if (val.size() == 8) memcpy(&delta, val.data(), 8);
                    auto it = kv.find(key);
                    if (it != kv.end() && it->second.size() == 8) memcpy(&cur, it->second.data(), 8);
                    cur += delta;
                    kv[key] = std::string((const char*)&cur, 8);
                    reply(c.fd, 0, kv[key]);
                } else {
                    reply(c.fd, 1, "");
                }
            }
        }
        // drop closed connections
        size_t w = 0;
        for (size_t i = 0; i < conns.size(); ++i)
            if (conns[i].fd >= 0) conns[w++] = conns[i];
        conns.resize(w);
    }
    for (auto& c : conns)
        if (c.fd >= 0) ::close(c.fd);
}

void TcpStore::request(uint8_t op, const std::string& key, const std::string& val,
                       std::string* out) {
    std::lock_guard<std::mutex> lock(client_mu_);
    std::string msg;
    msg.push_back((char)op);
    uint32_t klen = (uint32_t)key.size(), vlen = (uint32_t)val.size();
    msg.append((const char*)&klen, 4);
    msg.append(key);
    msg.append((const char*)&vlen, 4);
    msg.append(val);
    if (!send_all(client_fd_, msg.data(), msg.size()))
        throw std::runtime_error("TcpStore: connection to the store lost (send)");
    uint8_t status;
    uint32_t n;
    if (!recv_all(client_fd_, &status, 1, timeout_s_) || !recv_all(client_fd_, &n, 4, timeout_s_))
        throw std::runtime_error("TcpStore: timed out after " + std::to_string(timeout_s_) +
                                 " s waiting for key '" + key + "'");
    std::string body(n, '\0');
    if (n && !recv_all(client_fd_, &body[0], n, timeout_s_))
        throw std::runtime_error("TcpStore: truncated reply");
    if (status != 0) throw std::runtime_error("TcpStore: server rejected the request");
    if (out) *out = body;
}

void TcpStore::set(const std::string& key, const std::string& value) {
    request(1, key, value, nullptr);
}

std::string TcpStore::get(const std::string& key) {
    std::string out;
    request(2, key, "", &out);
    return out;
}

int64_t TcpStore::add(const std::string& key, int64_t delta) {
    std::string out;
    request(3, key, std::string((const char*)&delta, 8), &out);
    int64_t v = 0;
    if (out.size() == 8) memcpy(&v, out.data(), 8);
    return v;
}

void TcpStore::barrier(const std::string& name) {
    const std::string key = "__barrier__" + name;
    const auto t0 = Clock::now();
    int64_t arrived = add(key, 1);
    while (arrived < world_size_) {
        if (seconds_since(t0) > timeout_s_)
            throw std::runtime_error("TcpStore: barrier '" + name + "' timed out with " +
                                     std::to_string(arrived) + "/" + std::to_string(world_size_) +
                                     " ranks");
        ::usleep(2000);
        arrived = add(key, 0);
    }
}

}  // namespace dml

#include "singa/io/network.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstring>

#include "singa/utils/logging.h"

namespace singa {

namespace {
constexpr size_t kHeader = 13;  // type u8, id u32, meta u32, payload u32

void PutU32(std::string *s, uint32_t v) {
  unsigned char b[4] = {static_cast<unsigned char>(v >> 24),
                        static_cast<unsigned char>(v >> 16),
                        static_cast<unsigned char>(v >> 8),
                        static_cast<unsigned char>(v)};
  s->append(reinterpret_cast<char *>(b), 4);
}
uint32_t GetU32(const char *p) {
  auto *b = reinterpret_cast<const unsigned char *>(p);
  return (uint32_t(b[0]) << 24) | (uint32_t(b[1]) << 16) |
         (uint32_t(b[2]) << 8) | b[3];
}

std::string Frame(uint8_t type, uint32_t id, const std::string &meta,
                  const std::string &payload) {
  std::string s;
  s.reserve(kHeader + meta.size() + payload.size());
  s.push_back(static_cast<char>(type));
  PutU32(&s, id);
  PutU32(&s, static_cast<uint32_t>(meta.size()));
  PutU32(&s, static_cast<uint32_t>(payload.size()));
  s += meta;
  s += payload;
  return s;
}

bool WriteAll(int fd, const std::string &data) {
  size_t off = 0;
  while (off < data.size()) {
    ssize_t n = ::send(fd, data.data() + off, data.size() - off, MSG_NOSIGNAL);
    if (n > 0) {
      off += n;
    } else if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
      pollfd p{fd, POLLOUT, 0};
      if (::poll(&p, 1, EP_TIMEOUT * 1000) <= 0) return false;
    } else if (n < 0 && errno == EINTR) {
      continue;
    } else {
      return false;
    }
  }
  return true;
}

void SetNonBlocking(int fd) {
  int fl = fcntl(fd, F_GETFL, 0);
  fcntl(fd, F_SETFL, fl | O_NONBLOCK);
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}
}  // namespace

// ---- EndPoint ----------------------------------------------------------------------
EndPoint::~EndPoint() {
  for (Message *m : inbox_) delete m;
}

bool EndPoint::EnsureConnected() {
  if (wfd_ >= 0) return true;
  auto colon = key_.rfind(':');
  std::string host = key_.substr(0, colon), port = key_.substr(colon + 1);
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) != 0) return false;
  int fd = ::socket(AF_INET, SOCK_STREAM, 0);
  bool ok = fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0;
  freeaddrinfo(res);
  if (!ok) {
    if (fd >= 0) ::close(fd);
    return false;
  }
  SetNonBlocking(fd);
  // tell the peer which listening port identifies this side
  std::string me = std::to_string(nt_->port());
  if (!WriteAll(fd, Frame(MSG_HELLO, 0, me, ""))) {
    ::close(fd);
    return false;
  }
  wfd_ = fd;
  nt_->Watch(fd, this);
  return true;
}

int EndPoint::send(Message *msg) {
  uint32_t id;
  {
    std::lock_guard<std::mutex> lock(mtx_);
    id = next_id_++;
    unacked_.insert(id);
  }
  msg->id_ = id;
  std::string frame = Frame(MSG_DATA, id, msg->meta_, msg->payload_);
  for (int attempt = 0; attempt < MAX_RETRY_CNT; ++attempt) {
    {
      std::lock_guard<std::mutex> wlock(wmtx_);
      if (EnsureConnected() && WriteAll(wfd_, frame))
        return static_cast<int>(id);
      wfd_ = -1;  // broken or missing connection: drop it and try again
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(100 * (attempt + 1)));
  }
  std::lock_guard<std::mutex> lock(mtx_);
  unacked_.erase(id);
  return -1;
}

Message *EndPoint::recv(int timeout_sec) {
  std::unique_lock<std::mutex> lock(mtx_);
  auto ready = [&] { return !inbox_.empty(); };
  if (timeout_sec < 0) {
    cv_.wait(lock, ready);
  } else if (!cv_.wait_for(lock, std::chrono::seconds(timeout_sec), ready)) {
    return nullptr;
  }
  Message *m = inbox_.front();
  inbox_.pop_front();
  return m;
}

bool EndPoint::drain(int timeout_sec) {
  std::unique_lock<std::mutex> lock(mtx_);
  return cv_.wait_for(lock, std::chrono::seconds(timeout_sec),
                      [&] { return unacked_.empty(); });
}

int EndPoint::pending_acks() {
  std::lock_guard<std::mutex> lock(mtx_);
  return static_cast<int>(unacked_.size());
}

// ---- EndPointFactory ----------------------------------------------------------------
EndPointFactory::~EndPointFactory() {
  for (auto &kv : eps_) delete kv.second;
}

EndPoint *EndPointFactory::getOrCreateEp(const std::string &key) {
  std::lock_guard<std::mutex> lock(mtx_);
  auto it = eps_.find(key);
  if (it != eps_.end()) return it->second;
  EndPoint *ep = new EndPoint(nt_, key);
  eps_[key] = ep;
  return ep;
}

EndPoint *EndPointFactory::getEp(const std::string &host, int port) {
  // resolve to dotted form so that both sides derive the same key
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  std::string ip = host;
  if (getaddrinfo(host.c_str(), nullptr, &hints, &res) == 0 && res) {
    char buf[INET_ADDRSTRLEN];
    inet_ntop(AF_INET, &reinterpret_cast<sockaddr_in *>(res->ai_addr)->sin_addr,
              buf, sizeof(buf));
    ip = buf;
    freeaddrinfo(res);
  }
  return getOrCreateEp(ip + ":" + std::to_string(port));
}

EndPoint *EndPointFactory::getEp(const char *host) {
  std::string h(host);
  auto colon = h.rfind(':');
  if (colon != std::string::npos)
    return getEp(h.substr(0, colon), std::stoi(h.substr(colon + 1)));
  return getEp(h, nt_->port());
}

EndPoint *EndPointFactory::waitAny(int timeout_sec) {
  auto deadline = std::chrono::steady_clock::now() +
                  std::chrono::seconds(timeout_sec);
  std::unique_lock<std::mutex> lock(nt_->any_mtx_);
  while (true) {
    {
      std::lock_guard<std::mutex> l2(mtx_);
      for (auto &kv : eps_) {
        std::lock_guard<std::mutex> l3(kv.second->mtx_);
        if (!kv.second->inbox_.empty()) return kv.second;
      }
    }
    if (nt_->any_cv_.wait_until(lock, deadline) == std::cv_status::timeout)
      return nullptr;
  }
}

// ---- NetworkThread ------------------------------------------------------------------
NetworkThread::NetworkThread(int port) : port_(port) {
  epf_ = new EndPointFactory(this);
  listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
  CHECK_GE(listen_fd_, 0) << "socket() failed";
  int one = 1;
  setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_addr.s_addr = htonl(INADDR_ANY);
  addr.sin_port = htons(static_cast<uint16_t>(port));
  CHECK_EQ(::bind(listen_fd_, reinterpret_cast<sockaddr *>(&addr), sizeof(addr)), 0)
      << "cannot bind port " << port << ": " << std::strerror(errno);
  if (port == 0) {  // ephemeral port requested: find out which one we got
    socklen_t len = sizeof(addr);
    getsockname(listen_fd_, reinterpret_cast<sockaddr *>(&addr), &len);
    port_ = ntohs(addr.sin_port);
  }
  CHECK_EQ(::listen(listen_fd_, 128), 0);
  SetNonBlocking(listen_fd_);
  epoll_fd_ = epoll_create1(0);
  wake_fd_ = eventfd(0, EFD_NONBLOCK);
  epoll_event ev{};
  ev.events = EPOLLIN;
  ev.data.fd = listen_fd_;
  epoll_ctl(epoll_fd_, EPOLL_CTL_ADD, listen_fd_, &ev);
  ev.data.fd = wake_fd_;
  epoll_ctl(epoll_fd_, EPOLL_CTL_ADD, wake_fd_, &ev);
  thread_ = std::thread([this] { Loop(); });
  ack_thread_ = std::thread([this] { AckLoop(); });
}

void NetworkThread::AckLoop() {
  while (true) {
    std::pair<EndPoint *, uint32_t> item;
    {
      std::unique_lock<std::mutex> lock(ack_mtx_);
      ack_cv_.wait(lock, [&] { return stop_ || !ack_q_.empty(); });
      if (ack_q_.empty()) return;
      item = ack_q_.front();
      ack_q_.pop_front();
    }
    std::lock_guard<std::mutex> w(item.first->wmtx_);
    if (item.first->wfd_ >= 0)
      WriteAll(item.first->wfd_, Frame(MSG_ACK, item.second, "", ""));
  }
}

NetworkThread::~NetworkThread() {
  stop_ = true;
  uint64_t one = 1;
  (void)!::write(wake_fd_, &one, sizeof(one));
  if (thread_.joinable()) thread_.join();
  {
    std::lock_guard<std::mutex> lock(ack_mtx_);
    ack_cv_.notify_all();
  }
  if (ack_thread_.joinable()) ack_thread_.join();
  for (auto &kv : conns_) {
    ::close(kv.first);
    delete kv.second;
  }
  ::close(listen_fd_);
  ::close(epoll_fd_);
  ::close(wake_fd_);
  delete epf_;
}

void NetworkThread::Watch(int fd, EndPoint *ep) {
  Conn *c = new Conn();
  c->fd = fd;
  c->ep = ep;
  {
    std::lock_guard<std::mutex> lock(cmtx_);
    conns_[fd] = c;
  }
  epoll_event ev{};
  ev.events = EPOLLIN;
  ev.data.fd = fd;
  epoll_ctl(epoll_fd_, EPOLL_CTL_ADD, fd, &ev);
}

void NetworkThread::Accept() {
  while (true) {
    sockaddr_in peer{};
    socklen_t len = sizeof(peer);
    int fd = ::accept(listen_fd_, reinterpret_cast<sockaddr *>(&peer), &len);
    if (fd < 0) return;
    SetNonBlocking(fd);
    Watch(fd, nullptr);  // bound to an endpoint when its HELLO arrives
  }
}

void NetworkThread::Close(Conn *c) {
  epoll_ctl(epoll_fd_, EPOLL_CTL_DEL, c->fd, nullptr);
  if (c->ep) {
    std::lock_guard<std::mutex> w(c->ep->wmtx_);
    if (c->ep->wfd_ == c->fd) c->ep->wfd_ = -1;
  }
  ::close(c->fd);
  std::lock_guard<std::mutex> lock(cmtx_);
  conns_.erase(c->fd);
  delete c;
}

bool NetworkThread::Parse(Conn *c) {
  while (c->rbuf.size() >= kHeader) {
    uint8_t type = static_cast<uint8_t>(c->rbuf[0]);
    uint32_t id = GetU32(c->rbuf.data() + 1);
    uint32_t ms = GetU32(c->rbuf.data() + 5), ps = GetU32(c->rbuf.data() + 9);
    size_t total = kHeader + size_t(ms) + ps;
    // lengths come from the peer: a frame nobody sends (over 1 GiB) is a
    // protocol violation, not a reason to buffer until memory runs out
    if (total > (size_t(1) << 30)) return false;
    if (c->rbuf.size() < total) break;
    std::string meta = c->rbuf.substr(kHeader, ms);
    std::string payload = c->rbuf.substr(kHeader + ms, ps);
    c->rbuf.erase(0, total);
    if (type == MSG_HELLO) {
      sockaddr_in peer{};
      socklen_t len = sizeof(peer);
      getpeername(c->fd, reinterpret_cast<sockaddr *>(&peer), &len);
      char buf[INET_ADDRSTRLEN];
      inet_ntop(AF_INET, &peer.sin_addr, buf, sizeof(buf));
      c->ep = epf_->getOrCreateEp(std::string(buf) + ":" + meta);
      std::lock_guard<std::mutex> w(c->ep->wmtx_);
      if (c->ep->wfd_ < 0) c->ep->wfd_ = c->fd;  // reply over the same socket
    } else if (c->ep == nullptr) {
      return false;  // data before HELLO: protocol violation
    } else if (type == MSG_ACK) {
      std::lock_guard<std::mutex> lock(c->ep->mtx_);
      c->ep->unacked_.erase(id);
      c->ep->cv_.notify_all();
    } else if (type == MSG_DATA) {
      Message *m = new Message();
      m->id_ = id;
      m->meta_ = std::move(meta);
      m->payload_ = std::move(payload);
      {
        std::lock_guard<std::mutex> lock(c->ep->mtx_);
        c->ep->inbox_.push_back(m);
        c->ep->cv_.notify_all();
      }
      {
        std::lock_guard<std::mutex> lock(any_mtx_);
        any_cv_.notify_all();
      }
      {
        std::lock_guard<std::mutex> lock(ack_mtx_);
        ack_q_.emplace_back(c->ep, id);
        ack_cv_.notify_one();
      }
    } else {
      return false;
    }
  }
  return true;
}

void NetworkThread::OnReadable(Conn *c) {
  char buf[65536];
  while (true) {
    ssize_t n = ::recv(c->fd, buf, sizeof(buf), 0);
    if (n > 0) {
      c->rbuf.append(buf, n);
    } else if (n == 0) {
      Parse(c);
      Close(c);
      return;
    } else if (errno == EAGAIN || errno == EWOULDBLOCK) {
      break;
    } else if (errno == EINTR) {
      continue;
    } else {
      Close(c);
      return;
    }
  }
  if (!Parse(c)) Close(c);
}

void NetworkThread::Loop() {
  epoll_event events[64];
  while (!stop_) {
    int n = epoll_wait(epoll_fd_, events, 64, 500);
    for (int i = 0; i < n; ++i) {
      int fd = events[i].data.fd;
      if (fd == listen_fd_) {
        Accept();
      } else if (fd == wake_fd_) {
        uint64_t v;
        (void)!::read(wake_fd_, &v, sizeof(v));
      } else {
        Conn *c = nullptr;
        {
          std::lock_guard<std::mutex> lock(cmtx_);
          auto it = conns_.find(fd);
          if (it != conns_.end()) c = it->second;
        }
        if (c) OnReadable(c);
      }
    }
  }
}

}  // namespace singa

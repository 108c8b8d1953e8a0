// TCP message layer: a NetworkThread owns a listening socket and an epoll
// loop; EndPoints are bidirectional channels to peers identified by
// "ip:listening-port".  Every data message is acknowledged by the receiver;
// send() retries (reconnecting if needed) up to MAX_RETRY_CNT times and
// EndPoint::drain() waits for outstanding acknowledgements with a timeout.
//
// API parity: reference include/singa/io/network.h:40-168 (Message, EndPoint,
// EndPointFactory, NetworkThread; libev there, epoll here).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

namespace singa {

#define MSG_DATA 0
#define MSG_ACK 1
#define MSG_HELLO 2
#define MAX_RETRY_CNT 3
#define EP_TIMEOUT 5  // seconds

class Message {
 public:
  Message() = default;
  Message(const Message &) = default;
  Message(Message &&) = default;
  void setMetadata(const void *buf, int size) {
    meta_.assign(static_cast<const char *>(buf), size);
  }
  void setPayload(const void *buf, int size) {
    payload_.assign(static_cast<const char *>(buf), size);
  }
  std::size_t getMetadata(void **buf) {
    *buf = meta_.empty() ? nullptr : &meta_[0];
    return meta_.size();
  }
  std::size_t getPayload(void **buf) {
    *buf = payload_.empty() ? nullptr : &payload_[0];
    return payload_.size();
  }
  std::size_t getSize() const { return meta_.size() + payload_.size(); }
  uint32_t getId() const { return id_; }

 private:
  friend class EndPoint;
  friend class NetworkThread;
  uint8_t type_ = MSG_DATA;
  uint32_t id_ = 0;
  std::string meta_, payload_;
};

class NetworkThread;

class EndPoint {
 public:
  ~EndPoint();
  /// Queue-free blocking send; returns the message id, or -1 after
  /// MAX_RETRY_CNT failed attempts.  The message is copied onto the wire.
  int send(Message *msg);
  /// Next received data message (caller owns it); nullptr on timeout
  /// (timeout_sec < 0 waits forever).
  Message *recv(int timeout_sec = -1);
  /// Wait until every sent message has been acknowledged.
  bool drain(int timeout_sec = EP_TIMEOUT);
  const std::string &peer() const { return key_; }
  int pending_acks();

 private:
  friend class NetworkThread;
  friend class EndPointFactory;
  EndPoint(NetworkThread *nt, std::string key) : nt_(nt), key_(std::move(key)) {}
  bool EnsureConnected();
  NetworkThread *nt_;
  std::string key_;              // "ip:port" of the peer's listening socket
  int wfd_ = -1;                 // socket used for writing
  std::mutex wmtx_;              // serialises writers
  std::mutex mtx_;               // guards the members below
  std::condition_variable cv_;
  std::deque<Message *> inbox_;
  std::unordered_set<uint32_t> unacked_;
  uint32_t next_id_ = 1;
};

class EndPointFactory {
 public:
  explicit EndPointFactory(NetworkThread *nt) : nt_(nt) {}
  ~EndPointFactory();
  /// "host" (peer listens on this thread's own port number) or "host:port".
  EndPoint *getEp(const char *host);
  EndPoint *getEp(const std::string &host, int port);
  EndPoint *getOrCreateEp(const std::string &key);
  /// First endpoint that has an unread message, waiting up to timeout_sec.
  EndPoint *waitAny(int timeout_sec);

 private:
  friend class NetworkThread;
  NetworkThread *nt_;
  std::mutex mtx_;
  std::map<std::string, EndPoint *> eps_;
};

class NetworkThread {
 public:
  explicit NetworkThread(int port);
  ~NetworkThread();
  int port() const { return port_; }
  EndPointFactory *epf_;

 private:
  friend class EndPoint;
  friend class EndPointFactory;
  struct Conn {
    int fd = -1;
    EndPoint *ep = nullptr;
    std::string rbuf;
  };
  void Loop();
  void Accept();
  void OnReadable(Conn *c);
  bool Parse(Conn *c);
  void Watch(int fd, EndPoint *ep);
  void Close(Conn *c);
  int port_;
  int listen_fd_ = -1, epoll_fd_ = -1, wake_fd_ = -1;
  std::atomic<bool> stop_{false};
  std::thread thread_;
  std::mutex cmtx_;
  std::map<int, Conn *> conns_;
  std::mutex any_mtx_;
  std::condition_variable any_cv_;
  // acknowledgements are written by their own thread so that the epoll loop
  // never blocks on a full socket (it must keep draining reads)
  void AckLoop();
  std::thread ack_thread_;
  std::mutex ack_mtx_;
  std::condition_variable ack_cv_;
  std::deque<std::pair<EndPoint *, uint32_t>> ack_q_;
};

}  // namespace singa

// Named append-only sinks for metrics: each channel writes lines to a file
// under a global directory and / or to stderr.
//
// Behaviour parity: reference include/singa/utils/channel.h:35-81,
// src/utils/channel.cc (file destination enabled by default, stderr disabled,
// default file = <dir>/<channel name>).
#pragma once
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>

namespace singa_amd {

class Channel {
 public:
  explicit Channel(const std::string &name, const std::string &dir)
      : name_(name) {
    SetDestFilePath(dir.empty() ? name : dir + "/" + name);
  }
  const std::string &name() const { return name_; }
  void EnableDestStderr(bool on) { stderr_ = on; }
  void EnableDestFile(bool on) { file_ = on; }
  void SetDestFilePath(const std::string &path) {
    std::lock_guard<std::mutex> lk(mu_);
    if (os_.is_open()) os_.close();
    path_ = path;
  }
  void Send(const std::string &message) {
    std::lock_guard<std::mutex> lk(mu_);
    if (stderr_) std::cerr << message << std::endl;
    if (file_) {
      if (!os_.is_open()) os_.open(path_, std::ios::app);
      if (os_.is_open()) {
        os_ << message << std::endl;
        os_.flush();
      }
    }
  }

 private:
  std::string name_;
  std::string path_;
  bool stderr_ = false;
  bool file_ = true;
  std::ofstream os_;
  std::mutex mu_;
};

class ChannelManager {
 public:
  static ChannelManager &Get() {
    static ChannelManager m;
    return m;
  }
  void SetDefaultDir(const std::string &dir) {
    std::lock_guard<std::mutex> lk(mu_);
    dir_ = dir;
  }
  Channel *GetInstance(const std::string &name) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = chans_.find(name);
    if (it == chans_.end())
      it = chans_.emplace(name, std::make_unique<Channel>(name, dir_)).first;
    return it->second.get();
  }

 private:
  std::string dir_;
  std::map<std::string, std::unique_ptr<Channel>> chans_;
  std::mutex mu_;
};

inline void InitChannel(const std::string & /*argv0*/) { ChannelManager::Get(); }
inline void SetChannelDirectory(const std::string &path) {
  ChannelManager::Get().SetDefaultDir(path);
}
inline Channel *GetChannel(const std::string &name) {
  return ChannelManager::Get().GetInstance(name);
}

}  // namespace singa_amd

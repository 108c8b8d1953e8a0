// Key-value store files in the LMDB on-disk format (data version 1), read and
// written without the LMDB library.
//
// An environment is a directory holding `data.mdb`: a sequence of pages.
//   page header (16 bytes): u64 page number | u16 pad | u16 flags |
//                           u16 lower, u16 upper  (or u32 page count of an
//                           overflow run)
//   pages 0 and 1 are meta pages: header, then u32 magic 0xBEEFC0DE,
//     u32 version 1, u64 fixed address, u64 map size, two 48-byte database
//     records (free list, main), u64 last page, u64 transaction id.  The page
//     size lives in the pad word of the free-list record.  The meta page with
//     the larger transaction id is the current one.
//   database record: u32 pad | u16 flags | u16 depth | u64 branch pages |
//     u64 leaf pages | u64 overflow pages | u64 entries | u64 root page
//   branch / leaf page: after the header an array of u16 node offsets, sorted
//     by key; nodes are packed from the end of the page downwards.
//   node: u16 lo | u16 hi | u16 flags | u16 key size | key | data.
//     leaf: (lo | hi << 16) = data size; flag 1 = the data is the u64 number
//     of the first page of an overflow run holding the bytes.
//     branch: lo | hi << 16 | flags << 32 = child page; the first node of a
//     branch page has an empty key.
//   keys are ordered bytewise, a prefix before its extensions.
//
// The reader maps the file and walks the main database's tree with a cursor.
// The writer keeps the pending tuples in memory and, on Flush, rebuilds the
// whole tree bottom-up from the sorted tuples (sequential page fill, no free
// list), which is the cheapest way to produce a dataset file written once and
// read many times.  In append mode the existing entries are loaded first.
// A tuple written twice keeps the last value.
//
// API parity: reference src/io/lmdb_reader.cc:28-118 (Open / Read as
// first-then-next cursor / Count from the entry counter / SeekToFirst),
// src/io/lmdb_writer.cc:27-131 (Open creates the directory, Write buffers,
// Flush commits).  Files produced here were never opened by the LMDB library
// itself in this build's tests (the library is not installed): format parity
// unpinned; the tests parse the file with an independent reader instead.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "logging.h"

namespace singa_amd {
namespace lmdbfmt {

constexpr uint32_t kMagic = 0xBEEFC0DE;
constexpr uint32_t kVersion = 1;
constexpr uint64_t kInvalidPage = ~uint64_t(0);
constexpr size_t kPageHeader = 16;
constexpr size_t kNodeHeader = 8;
constexpr size_t kMaxKey = 511;
constexpr uint16_t kBranch = 0x01, kLeaf = 0x02, kOverflow = 0x04,
                   kMeta = 0x08;
constexpr uint16_t kBigData = 0x01;

struct Db {
  uint32_t pad = 0;
  uint16_t flags = 0;
  uint16_t depth = 0;
  uint64_t branch_pages = 0, leaf_pages = 0, overflow_pages = 0, entries = 0;
  uint64_t root = kInvalidPage;
};

struct Meta {
  uint32_t magic = kMagic, version = kVersion;
  uint64_t address = 0, mapsize = 0;
  Db free_db, main_db;
  uint64_t last_page = 1, txnid = 0;
};

template <typename T>
inline T Load(const uint8_t *p) {
  T v;
  std::memcpy(&v, p, sizeof(T));
  return v;
}
template <typename T>
inline void Store(uint8_t *p, T v) {
  std::memcpy(p, &v, sizeof(T));
}

inline Db LoadDb(const uint8_t *p) {
  Db d;
  d.pad = Load<uint32_t>(p);
  d.flags = Load<uint16_t>(p + 4);
  d.depth = Load<uint16_t>(p + 6);
  d.branch_pages = Load<uint64_t>(p + 8);
  d.leaf_pages = Load<uint64_t>(p + 16);
  d.overflow_pages = Load<uint64_t>(p + 24);
  d.entries = Load<uint64_t>(p + 32);
  d.root = Load<uint64_t>(p + 40);
  return d;
}
inline void StoreDb(uint8_t *p, const Db &d) {
  Store<uint32_t>(p, d.pad);
  Store<uint16_t>(p + 4, d.flags);
  Store<uint16_t>(p + 6, d.depth);
  Store<uint64_t>(p + 8, d.branch_pages);
  Store<uint64_t>(p + 16, d.leaf_pages);
  Store<uint64_t>(p + 24, d.overflow_pages);
  Store<uint64_t>(p + 32, d.entries);
  Store<uint64_t>(p + 40, d.root);
}

/// Key order of the default comparison: bytewise, a prefix sorts first.
struct KeyLess {
  bool operator()(const std::string &a, const std::string &b) const {
    size_t n = std::min(a.size(), b.size());
    int c = std::memcmp(a.data(), b.data(), n);
    return c != 0 ? c < 0 : a.size() < b.size();
  }
};

}  // namespace lmdbfmt

class LMDBReader {
 public:
  ~LMDBReader() { Close(); }

  bool Open(const std::string &path) {
    using namespace lmdbfmt;
    Close();
    path_ = path;
    std::string file = path + "/data.mdb";
    fd_ = ::open(file.c_str(), O_RDONLY);
    if (fd_ < 0) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 2 * 512) {
      Close();
      return false;
    }
    size_ = static_cast<size_t>(st.st_size);
    void *m = mmap(nullptr, size_, PROT_READ, MAP_SHARED, fd_, 0);
    if (m == MAP_FAILED) {
      Close();
      return false;
    }
    base_ = static_cast<const uint8_t *>(m);
    // Meta page 0 tells the page size; meta page 1 sits one page further.
    Meta m0;
    if (!ParseMeta(base_, &m0)) {
      Close();
      return false;
    }
    psize_ = m0.free_db.pad;
    if (psize_ < 512 || (psize_ & (psize_ - 1)) || 2 * psize_ > size_) {
      Close();
      return false;
    }
    Meta m1;
    meta_ = m0;
    if (ParseMeta(base_ + psize_, &m1) && m1.txnid > m0.txnid) meta_ = m1;
    npages_ = size_ / psize_;
    SeekToFirst();
    return true;
  }

  void Close() {
    if (base_ != nullptr) munmap(const_cast<uint8_t *>(base_), size_);
    base_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
    stack_.clear();
  }

  /// The first call after Open / SeekToFirst yields the first tuple, every
  /// later call the next one; false past the last tuple.
  bool Read(std::string *key, std::string *value) {
    SA_CHECK(base_ != nullptr) << "LMDB environment not open";
    if (first_) {
      first_ = false;
      valid_ = Descend(meta_.main_db.root);
    } else if (valid_) {
      valid_ = Advance();
    }
    if (!valid_) return false;
    Current(key, value);
    return true;
  }

  int Count() { return static_cast<int>(meta_.main_db.entries); }

  void SeekToFirst() {
    stack_.clear();
    first_ = true;
    valid_ = false;
  }

  bool valid() const { return valid_; }
  const std::string &path() const { return path_; }
  size_t page_size() const { return psize_; }
  int depth() const { return meta_.main_db.depth; }

 private:
  struct Pos {
    uint64_t page;
    unsigned idx;
  };

  static bool ParseMeta(const uint8_t *page, lmdbfmt::Meta *m) {
    using namespace lmdbfmt;
    if (!(Load<uint16_t>(page + 10) & kMeta)) return false;
    const uint8_t *p = page + kPageHeader;
    m->magic = Load<uint32_t>(p);
    m->version = Load<uint32_t>(p + 4);
    if (m->magic != kMagic || m->version != kVersion) return false;
    m->address = Load<uint64_t>(p + 8);
    m->mapsize = Load<uint64_t>(p + 16);
    m->free_db = LoadDb(p + 24);
    m->main_db = LoadDb(p + 72);
    m->last_page = Load<uint64_t>(p + 120);
    m->txnid = Load<uint64_t>(p + 128);
    return true;
  }

  const uint8_t *Page(uint64_t no) const {
    SA_CHECK_LT(no, npages_) << "page number outside the file";
    return base_ + no * psize_;
  }
  unsigned NumKeys(const uint8_t *pg) const {
    unsigned lower = lmdbfmt::Load<uint16_t>(pg + 12);
    SA_CHECK_GE(lower, lmdbfmt::kPageHeader);
    return (lower - lmdbfmt::kPageHeader) >> 1;
  }
  const uint8_t *Node(const uint8_t *pg, unsigned i) const {
    unsigned off = lmdbfmt::Load<uint16_t>(pg + lmdbfmt::kPageHeader + 2 * i);
    SA_CHECK_LE(off + lmdbfmt::kNodeHeader, psize_) << "node outside its page";
    return pg + off;
  }

  /// Walks from `page` down the leftmost path to a leaf.
  bool Descend(uint64_t page) {
    using namespace lmdbfmt;
    if (page == kInvalidPage) return false;
    for (int guard = 0; guard < 64; ++guard) {
      const uint8_t *pg = Page(page);
      uint16_t flags = Load<uint16_t>(pg + 10);
      if (NumKeys(pg) == 0) return false;
      stack_.push_back({page, 0});
      if (flags & kLeaf) return true;
      SA_CHECK(flags & kBranch) << "unexpected page type in the tree";
      page = ChildOf(Node(pg, 0));
    }
    SA_LOG(FATAL) << "tree deeper than 64 levels";
    return false;
  }

  static uint64_t ChildOf(const uint8_t *node) {
    using namespace lmdbfmt;
    return uint64_t(Load<uint16_t>(node)) |
           (uint64_t(Load<uint16_t>(node + 2)) << 16) |
           (uint64_t(Load<uint16_t>(node + 4)) << 32);
  }

  bool Advance() {
    while (!stack_.empty()) {
      Pos &top = stack_.back();
      const uint8_t *pg = Page(top.page);
      if (top.idx + 1 < NumKeys(pg)) {
        ++top.idx;
        if (lmdbfmt::Load<uint16_t>(pg + 10) & lmdbfmt::kLeaf) return true;
        return Descend(ChildOf(Node(pg, top.idx)));
      }
      stack_.pop_back();
    }
    return false;
  }

  void Current(std::string *key, std::string *value) const {
    using namespace lmdbfmt;
    const Pos &top = stack_.back();
    const uint8_t *pg = Page(top.page);
    const uint8_t *node = Node(pg, top.idx);
    size_t dsize = size_t(Load<uint16_t>(node)) |
                   (size_t(Load<uint16_t>(node + 2)) << 16);
    uint16_t flags = Load<uint16_t>(node + 4);
    size_t ksize = Load<uint16_t>(node + 6);
    const uint8_t *kp = node + kNodeHeader;
    SA_CHECK_LE(size_t(kp - pg) + ksize, psize_);
    key->assign(reinterpret_cast<const char *>(kp), ksize);
    if (flags & kBigData) {
      uint64_t ov = Load<uint64_t>(kp + ksize);
      const uint8_t *op = Page(ov);
      SA_CHECK(Load<uint16_t>(op + 10) & kOverflow);
      SA_CHECK_LE(ov * psize_ + kPageHeader + dsize, size_)
          << "overflow run outside the file";
      value->assign(reinterpret_cast<const char *>(op + kPageHeader), dsize);
    } else {
      SA_CHECK_LE(size_t(kp - pg) + ksize + dsize, psize_);
      value->assign(reinterpret_cast<const char *>(kp + ksize), dsize);
    }
  }

  std::string path_;
  int fd_ = -1;
  const uint8_t *base_ = nullptr;
  size_t size_ = 0, psize_ = 0, npages_ = 0;
  lmdbfmt::Meta meta_;
  std::vector<Pos> stack_;
  bool first_ = true, valid_ = false;
};

class LMDBWriter {
 public:
  ~LMDBWriter() { Close(); }

  /// `append` keeps the tuples an existing environment holds.
  bool Open(const std::string &path, bool append = false,
            size_t page_size = 4096) {
    SA_CHECK(page_size >= 512 && page_size <= 32768 &&
             !(page_size & (page_size - 1)))
        << "page size must be a power of two in [512, 32768]";
    path_ = path;
    psize_ = page_size;
    open_ = false;
    entries_.clear();
    dirty_ = false;
    struct stat st;
    bool exists = stat(path.c_str(), &st) == 0;
    if (!append) {
      if (!exists && mkdir(path.c_str(), 0744) != 0) return false;
      dirty_ = true;  // an empty environment is still written out
    } else {
      if (!exists) return false;
      LMDBReader r;
      if (r.Open(path)) {
        psize_ = r.page_size();
        std::string k, v;
        while (r.Read(&k, &v)) entries_[k] = v;
      }
    }
    open_ = true;
    return true;
  }

  bool Write(const std::string &key, const std::string &value) {
    SA_CHECK(open_) << "LMDB environment not open";
    SA_CHECK(!key.empty()) << "Key is an empty string!";
    SA_CHECK_LE(key.size(), lmdbfmt::kMaxKey) << "key longer than 511 bytes";
    SA_CHECK_LT(value.size(), size_t(1) << 32) << "value of 4 GiB or more";
    entries_[key] = value;
    dirty_ = true;
    return true;
  }

  /// Commits every tuple written so far: the file is rebuilt as a whole.
  void Flush() {
    if (!open_ || !dirty_) return;
    Build();
    dirty_ = false;
  }

  void Close() {
    if (open_) Flush();
    open_ = false;
    entries_.clear();
  }

  const std::string &path() const { return path_; }

 private:
  using Entry = std::pair<const std::string, std::string>;

  size_t NodeMax() const {
    return (((psize_ - lmdbfmt::kPageHeader) / 2) & ~size_t(1)) - 2;
  }
  static size_t Even(size_t n) { return (n + 1) & ~size_t(1); }

  struct PageBuilder {
    std::vector<uint8_t> bytes;
    size_t lower, upper;
    PageBuilder(size_t psize, uint64_t no, uint16_t flags)
        : bytes(psize, 0), lower(lmdbfmt::kPageHeader), upper(psize) {
      lmdbfmt::Store<uint64_t>(bytes.data(), no);
      lmdbfmt::Store<uint16_t>(bytes.data() + 10, flags);
    }
    bool Fits(size_t node_bytes) const {
      return lower + 2 + node_bytes <= upper;
    }
    uint8_t *Add(size_t node_bytes) {
      upper -= node_bytes;
      lmdbfmt::Store<uint16_t>(bytes.data() + lower,
                               static_cast<uint16_t>(upper));
      lower += 2;
      return bytes.data() + upper;
    }
    void Seal() {
      lmdbfmt::Store<uint16_t>(bytes.data() + 12, static_cast<uint16_t>(lower));
      lmdbfmt::Store<uint16_t>(bytes.data() + 14, static_cast<uint16_t>(upper));
    }
  };

  struct Child {
    uint64_t page;
    std::string first_key;
  };

  void Build() {
    using namespace lmdbfmt;
    std::string file = path_ + "/data.mdb";
    std::string tmp = file + ".tmp";
    std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
    SA_CHECK(out.is_open()) << "cannot write " << tmp;
    std::vector<uint8_t> blank(2 * psize_, 0);
    out.write(reinterpret_cast<const char *>(blank.data()),
              static_cast<std::streamsize>(blank.size()));
    uint64_t next = 2;
    Db db;
    const size_t node_max = NodeMax();

    // Leaf level: sequential fill; big values go to overflow runs written
    // right behind the leaf that refers to them.
    std::vector<Child> level;
    std::vector<std::pair<uint64_t, const std::string *>> pending_ov;
    PageBuilder *leaf = nullptr;
    std::string leaf_first;
    auto close_leaf = [&]() {
      if (leaf == nullptr) return;
      leaf->Seal();
      out.write(reinterpret_cast<const char *>(leaf->bytes.data()),
                static_cast<std::streamsize>(psize_));
      level.push_back({Load<uint64_t>(leaf->bytes.data()), leaf_first});
      ++db.leaf_pages;
      delete leaf;
      leaf = nullptr;
      for (auto &ov : pending_ov) {
        size_t pages = (kPageHeader - 1 + ov.second->size()) / psize_ + 1;
        std::vector<uint8_t> run(pages * psize_, 0);
        Store<uint64_t>(run.data(), ov.first);
        Store<uint16_t>(run.data() + 10, kOverflow);
        Store<uint32_t>(run.data() + 12, static_cast<uint32_t>(pages));
        std::memcpy(run.data() + kPageHeader, ov.second->data(),
                    ov.second->size());
        out.write(reinterpret_cast<const char *>(run.data()),
                  static_cast<std::streamsize>(run.size()));
        db.overflow_pages += pages;
      }
      pending_ov.clear();
    };
    // Page numbers are handed out in file order: a leaf takes `next`, its
    // overflow runs take the numbers after it, reserved while it fills.
    uint64_t reserved = 0;
    for (const Entry &e : entries_) {
      const std::string &k = e.first, &v = e.second;
      bool big = kNodeHeader + k.size() + v.size() > node_max;
      size_t nbytes = Even(kNodeHeader + k.size() + (big ? 8 : v.size()));
      if (leaf != nullptr && !leaf->Fits(nbytes)) {
        close_leaf();
        next += reserved;
        reserved = 0;
      }
      if (leaf == nullptr) {
        leaf = new PageBuilder(psize_, next++, kLeaf);
        leaf_first = k;
      }
      uint8_t *n = leaf->Add(nbytes);
      Store<uint16_t>(n, static_cast<uint16_t>(v.size() & 0xffff));
      Store<uint16_t>(n + 2, static_cast<uint16_t>(v.size() >> 16));
      Store<uint16_t>(n + 4, big ? kBigData : 0);
      Store<uint16_t>(n + 6, static_cast<uint16_t>(k.size()));
      std::memcpy(n + kNodeHeader, k.data(), k.size());
      if (big) {
        size_t pages = (kPageHeader - 1 + v.size()) / psize_ + 1;
        uint64_t ov = next + reserved;
        reserved += pages;
        Store<uint64_t>(n + kNodeHeader + k.size(), ov);
        pending_ov.push_back({ov, &v});
      } else {
        std::memcpy(n + kNodeHeader + k.size(), v.data(), v.size());
      }
      ++db.entries;
    }
    close_leaf();
    next += reserved;
    if (!level.empty()) db.depth = 1;

    // Branch levels until a single root is left.
    while (level.size() > 1) {
      // Greedy grouping of the children, then the last group borrows one
      // child if it ended up alone (a branch page needs two children).
      std::vector<std::vector<size_t>> groups(1);
      size_t used = kPageHeader;
      for (size_t i = 0; i < level.size(); ++i) {
        size_t ks = groups.back().empty() ? 0 : level[i].first_key.size();
        size_t need = 2 + Even(kNodeHeader + ks);
        if (used + need > psize_) {
          groups.emplace_back();
          used = kPageHeader;
          need = 2 + Even(kNodeHeader);
        }
        groups.back().push_back(i);
        used += need;
      }
      if (groups.size() > 1 && groups.back().size() == 1) {
        std::vector<size_t> &prev = groups[groups.size() - 2];
        groups.back().insert(groups.back().begin(), prev.back());
        prev.pop_back();
      }
      std::vector<Child> up;
      for (const std::vector<size_t> &g : groups) {
        PageBuilder pg(psize_, next++, kBranch);
        for (size_t j = 0; j < g.size(); ++j) {
          const Child &c = level[g[j]];
          size_t ks = j == 0 ? 0 : c.first_key.size();
          uint8_t *n = pg.Add(Even(kNodeHeader + ks));
          Store<uint16_t>(n, static_cast<uint16_t>(c.page & 0xffff));
          Store<uint16_t>(n + 2, static_cast<uint16_t>((c.page >> 16) & 0xffff));
          Store<uint16_t>(n + 4, static_cast<uint16_t>((c.page >> 32) & 0xffff));
          Store<uint16_t>(n + 6, static_cast<uint16_t>(ks));
          if (ks) std::memcpy(n + kNodeHeader, c.first_key.data(), ks);
        }
        pg.Seal();
        out.write(reinterpret_cast<const char *>(pg.bytes.data()),
                  static_cast<std::streamsize>(psize_));
        up.push_back({Load<uint64_t>(pg.bytes.data()), level[g[0]].first_key});
        ++db.branch_pages;
      }
      level.swap(up);
      ++db.depth;
    }
    if (!level.empty()) db.root = level[0].page;

    // Meta pages: 0 describes the empty environment (transaction 0), 1 the
    // tree just written (transaction 1), so readers pick page 1.
    Meta m;
    m.mapsize = std::max<uint64_t>(uint64_t(1) << 20, next * psize_);
    m.free_db.pad = static_cast<uint32_t>(psize_);
    std::vector<uint8_t> metas(2 * psize_, 0);
    for (int i = 0; i < 2; ++i) {
      uint8_t *pg = metas.data() + i * psize_;
      Store<uint64_t>(pg, uint64_t(i));
      Store<uint16_t>(pg + 10, kMeta);
      uint8_t *p = pg + kPageHeader;
      Store<uint32_t>(p, m.magic);
      Store<uint32_t>(p + 4, m.version);
      Store<uint64_t>(p + 8, m.address);
      Store<uint64_t>(p + 16, m.mapsize);
      StoreDb(p + 24, m.free_db);
      StoreDb(p + 72, i == 1 ? db : Db());
      Store<uint64_t>(p + 120, i == 1 ? next - 1 : uint64_t(1));
      Store<uint64_t>(p + 128, uint64_t(i));
    }
    out.seekp(0);
    out.write(reinterpret_cast<const char *>(metas.data()),
              static_cast<std::streamsize>(metas.size()));
    out.close();
    SA_CHECK(out.good()) << "write error on " << tmp;
    SA_CHECK_EQ(std::rename(tmp.c_str(), file.c_str()), 0)
        << "cannot replace " << file;
  }

  std::string path_;
  size_t psize_ = 4096;
  bool open_ = false, dirty_ = false;
  std::map<std::string, std::string, lmdbfmt::KeyLess> entries_;
};

}  // namespace singa_amd

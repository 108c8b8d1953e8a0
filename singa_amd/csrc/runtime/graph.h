// Computational-graph recorder and scheduler.
//
// Tensor operations are submitted as closures together with the blocks they
// read and write.  The recorder derives RAW / WAR / WAW dependencies between
// closures, classifies every block (input / parameter / intermediate / end),
// computes a replay plan and, for every plan step, the list of intermediate
// blocks whose storage can be released after that step.  A plan can be
// replayed as a whole or in contiguous ranges; ranges are what the HIP device
// captures into hipGraphs, with the non-capturable steps (collectives, host
// syncs) executed between the captured ranges.
//
// Behaviour parity: reference include/singa/core/scheduler.h:57-222 and
// src/core/scheduler/scheduler.cc:406-688 (edge construction, sync ops,
// breadth-first plan, free lists keyed on graph_ref >= Block::ref_count()).
#pragma once
#include <chrono>
#include <cstddef>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "block.h"

namespace singa_amd {

class Node;
class Edge;
class BlkInfo;
class Graph;

using OpFunc = std::function<void()>;
using NodeVec = std::vector<Node *>;
using EdgeVec = std::vector<Edge *>;
using BlockVec = std::vector<Block *>;

enum class BlockType : int { kUnknown = 0, kInput, kParam, kInter, kEnd };

class Node {
 public:
  Node(int id, OpFunc &&op, std::string name)
      : id_(id), op_(std::move(op)), name_(std::move(name)) {}
  int id() const { return id_; }
  const std::string &op_name() const { return name_; }
  const EdgeVec &in_edges() const { return in_; }
  const EdgeVec &out_edges() const { return out_; }
  float time_elapsed() const { return elapsed_; }
  void time_elapsed_inc(float t) { elapsed_ += t; }
  void Run() const { op_(); }

 private:
  friend class Graph;
  int id_;
  OpFunc op_;
  std::string name_;
  EdgeVec in_, out_;
  float elapsed_ = 0.f;
};

class Edge {
 public:
  Edge(int id, Block *blk, Node *src, Node *dst)
      : id_(id), blk_(blk), src_(src), dst_(dst) {}
  int id() const { return id_; }
  Block *block() const { return blk_; }
  Node *src_node() const { return src_; }
  Node *dst_node() const { return dst_; }

 private:
  friend class Graph;
  int id_;
  Block *blk_;
  Node *src_;
  Node *dst_;
};

class BlkInfo {
 public:
  BlkInfo(int id, Block *blk, BlockType type)
      : id_(id), blk_(blk), type_(type) {}
  int id() const { return id_; }
  Block *block() const { return blk_; }
  BlockType type() const { return type_; }
  int graph_ref() const { return graph_ref_; }
  Edge *write_edge() const { return write_edge_; }
  const NodeVec &used_nodes() const { return used_; }

 private:
  friend class Graph;
  int id_;
  Block *blk_;
  BlockType type_;
  int graph_ref_ = 0;
  Edge *write_edge_ = nullptr;  // edge leaving the most recent writer
  NodeVec used_;                // nodes touching the block, in plan order
};

/// Device-specific behaviour the scheduler needs while replaying.
struct ExecHooks {
  /// run one node with device-side timing around it (events or wall clock)
  std::function<void(Node &)> timed_exec;
  /// block the host until the device is idle
  std::function<void()> sync;
  /// seconds measured by the last timed_exec of this node
  std::function<float(Node &)> collect;
};

class Graph {
 public:
  Graph() = default;
  ~Graph();
  Graph(const Graph &) = delete;
  Graph &operator=(const Graph &) = delete;

  void Reset();
  void AddOperation(OpFunc &&op, const BlockVec &reads, const BlockVec &writes,
                    const std::string &name = "no_name");

  /// Replay in dependency (breadth-first) order.
  void RunGraph();
  /// Replay in submission order.
  void RunInSerial();

  // --- ranged replay (used for segmented hipGraph capture) -----------------
  /// Make sure the plan for the requested mode exists; returns its length.
  size_t Prepare(bool serial);
  /// Execute plan steps [first, last) and release their dead blocks.
  void RunRange(size_t first, size_t last);
  /// Mark one full iteration as done (advances the profiling counter).
  void FinishIteration() { ++iteration_; }
  /// While set, every node goes through the timed_exec hook whatever the
  /// iteration count (a device that captures the plan into a hipGraph puts
  /// its time stamps into the capture this way).
  void set_force_profile(bool on) { force_profile_ = on; }
  /// Times measured by the device itself for replays this graph did not
  /// execute node by node.
  void AddNodeTime(int id, float sec) { nodes_.at(id)->elapsed_ += sec; }
  void AddTimeElapsed(float sec) { time_elapsed_ += sec; }

  std::string Debug() const;
  std::string TimeProfiling() const;

  // --- configuration ---------------------------------------------------------
  void set_hooks(ExecHooks hooks) { hooks_ = std::move(hooks); }
  void set_verbosity(int v) { verbosity_ = v; }
  int verbosity() const { return verbosity_; }
  void set_skip_iteration(int n) { skip_iteration_ = n; }
  int skip_iteration() const { return skip_iteration_; }

  // --- introspection ---------------------------------------------------------
  const NodeVec &nodes() const { return nodes_; }
  const EdgeVec &edges() const { return edges_; }
  const std::vector<BlkInfo *> &block_infos() const { return infos_; }
  BlkInfo *block(Block *blk) const;
  const BlockVec &leaf_blocks() const { return leaf_order_; }
  bool dirty() const { return dirty_; }
  const NodeVec &begin_nodes() const { return begin_nodes_; }
  const std::vector<NodeVec> &next_nodes() const { return next_nodes_; }
  const std::vector<BlockVec> &free_blocks() const { return free_blocks_; }
  const std::vector<int> &plan() const { return plan_; }
  int iteration() const { return iteration_; }
  float time_elapsed() const { return time_elapsed_; }

 private:
  using Clock = std::chrono::high_resolution_clock;
  void AddSyncOp(OpFunc &&op, const std::string &name);
  void Analyze();
  void AnalyzeNodes();
  void AnalyzeEdges();
  void ExecNode(Node *n);
  bool profiling_now() const {
    return force_profile_ ||
           (verbosity_ > 0 && iteration_ >= skip_iteration_);
  }
  void RunPlan();
  BlkInfo *NewInfo(Block *blk, BlockType t);
  void LeafInsert(Block *b);
  void LeafErase(Block *b);

  NodeVec nodes_;
  EdgeVec edges_;      // edges visible to introspection (matches the reference)
  EdgeVec aux_edges_;  // ordering-only (write-after-read) edges
  std::unordered_map<Block *, BlkInfo *> blocks_;
  std::vector<BlkInfo *> infos_;  // same objects, creation order
  BlockVec leaf_order_;
  std::unordered_set<Block *> leaf_set_;

  bool dirty_ = false;
  bool in_serial_ = false;
  NodeVec begin_nodes_;
  std::vector<NodeVec> next_nodes_;
  std::vector<BlockVec> free_blocks_;
  std::vector<int> plan_;

  ExecHooks hooks_;
  int verbosity_ = 0;
  int skip_iteration_ = 5;
  int iteration_ = 0;
  float time_elapsed_ = 0.f;
  bool force_profile_ = false;
};

}  // namespace singa_amd

#include "graph.h"

#include <algorithm>
#include <deque>
#include <sstream>

namespace singa_amd {

Graph::~Graph() { Reset(); }

void Graph::Reset() {
  for (auto *n : nodes_) delete n;
  for (auto *e : edges_) delete e;
  for (auto *e : aux_edges_) delete e;
  for (auto *i : infos_) delete i;
  nodes_.clear();
  edges_.clear();
  aux_edges_.clear();
  infos_.clear();
  blocks_.clear();
  leaf_order_.clear();
  leaf_set_.clear();
  begin_nodes_.clear();
  next_nodes_.clear();
  free_blocks_.clear();
  plan_.clear();
  iteration_ = 0;
  time_elapsed_ = 0.f;
  dirty_ = false;
  in_serial_ = false;
}

BlkInfo *Graph::block(Block *blk) const {
  auto it = blocks_.find(blk);
  return it == blocks_.end() ? nullptr : it->second;
}

BlkInfo *Graph::NewInfo(Block *blk, BlockType t) {
  auto *info = new BlkInfo(static_cast<int>(infos_.size()), blk, t);
  infos_.push_back(info);
  blocks_[blk] = info;
  return info;
}

void Graph::LeafInsert(Block *b) {
  if (leaf_set_.insert(b).second) leaf_order_.push_back(b);
}

void Graph::LeafErase(Block *b) {
  if (leaf_set_.erase(b))
    leaf_order_.erase(std::find(leaf_order_.begin(), leaf_order_.end(), b));
}

void Graph::AddOperation(OpFunc &&op, const BlockVec &reads,
                         const BlockVec &writes, const std::string &name) {
  dirty_ = true;
  if (reads.empty() && writes.empty()) {
    AddSyncOp(std::move(op), name);
    return;
  }
  auto *node = new Node(static_cast<int>(nodes_.size()), std::move(op), name);

  // Read-after-write: connect the node to whoever last produced each input.
  for (Block *blk : reads) {
    LeafErase(blk);
    Node *producer = nullptr;
    BlkInfo *info = block(blk);
    if (!info) {
      info = NewInfo(blk, BlockType::kInput);
    } else {
      if (info->type_ == BlockType::kEnd) info->type_ = BlockType::kInter;
      if (Edge *w = info->write_edge_) {
        if (!w->dst_) {  // the dangling producer edge gets its first consumer
          w->dst_ = node;
          node->in_.push_back(w);
          info->graph_ref_ += 1;
          continue;
        }
        producer = w->src_;
      }
    }
    auto *e = new Edge(static_cast<int>(edges_.size()), blk, producer, node);
    info->graph_ref_ += 1;
    if (producer) producer->out_.push_back(e);
    node->in_.push_back(e);
    edges_.push_back(e);
  }

  // Write-after-write / write-after-read: order against the previous version.
  for (Block *blk : writes) {
    LeafInsert(blk);
    BlkInfo *info = block(blk);
    if (!info) {
      info = NewInfo(blk, BlockType::kEnd);
    } else {
      if (info->type_ == BlockType::kInput) info->type_ = BlockType::kParam;
      if (Edge *w = info->write_edge_) {
        if (!w->dst_) {
          w->dst_ = node;
          node->in_.push_back(w);
        } else {
          // every consumer of the previous version must finish first
          EdgeVec prev_out = w->src_->out_;
          for (Edge *o : prev_out) {
            if (o->blk_ == blk && o->dst_ != node) {
              auto *e =
                  new Edge(static_cast<int>(edges_.size()), blk, o->dst_, node);
              o->dst_->out_.push_back(e);
              node->in_.push_back(e);
              aux_edges_.push_back(e);
            }
          }
        }
      }
    }
    auto *e = new Edge(static_cast<int>(edges_.size()), blk, node, nullptr);
    info->write_edge_ = e;
    info->graph_ref_ += 1;
    node->out_.push_back(e);
    edges_.push_back(e);
  }
  nodes_.push_back(node);
}

void Graph::AddSyncOp(OpFunc &&op, const std::string &name) {
  auto *node = new Node(static_cast<int>(nodes_.size()), std::move(op), name);
  // A sync op depends on every block that is currently a leaf, and becomes
  // the producer of all of them so later ops are ordered behind it.
  for (Block *blk : leaf_order_) {
    BlkInfo *info = blocks_[blk];
    if (info->type_ == BlockType::kEnd) info->type_ = BlockType::kInter;
    Edge *w = info->write_edge_;
    Edge *in = nullptr;
    if (!w->dst_) {
      w->dst_ = node;
      in = w;
    } else {
      in = new Edge(static_cast<int>(edges_.size()), blk, w->src_, node);
      w->src_->out_.push_back(in);
      edges_.push_back(in);
    }
    node->in_.push_back(in);
    // pass-through edge: not a real use, so graph_ref is left untouched
    auto *out = new Edge(static_cast<int>(edges_.size()), blk, node, nullptr);
    info->write_edge_ = out;
    node->out_.push_back(out);
    edges_.push_back(out);
  }
  nodes_.push_back(node);
}

void Graph::Analyze() {
  begin_nodes_.clear();
  next_nodes_.assign(nodes_.size(), NodeVec());
  free_blocks_.assign(nodes_.size(), BlockVec());
  plan_.clear();
  for (auto *info : infos_) info->used_.clear();
  AnalyzeNodes();
  AnalyzeEdges();
  dirty_ = false;
}

static void MarkUse(Node *n, std::unordered_map<Block *, BlkInfo *> &blocks,
                    std::vector<BlkInfo *> &touched) {
  touched.clear();
  auto add = [&](Edge *e) {
    BlkInfo *info = blocks[e->block()];
    if (std::find(touched.begin(), touched.end(), info) == touched.end())
      touched.push_back(info);
  };
  for (Edge *e : n->in_edges()) add(e);
  for (Edge *e : n->out_edges()) add(e);
}

void Graph::AnalyzeNodes() {
  if (nodes_.empty()) return;
  std::vector<BlkInfo *> touched;
  if (in_serial_) {
    begin_nodes_.push_back(nodes_[0]);
    for (size_t i = 0; i < nodes_.size(); ++i) {
      if (i + 1 < nodes_.size()) next_nodes_[i].push_back(nodes_[i + 1]);
      plan_.push_back(static_cast<int>(i));
      MarkUse(nodes_[i], blocks_, touched);
      for (auto *info : touched) info->used_.push_back(nodes_[i]);
    }
    return;
  }
  // pending[i] = number of in-edges that still wait for their producer
  std::vector<int> pending(nodes_.size());
  for (size_t i = 0; i < nodes_.size(); ++i)
    pending[i] = static_cast<int>(nodes_[i]->in_.size());
  for (Edge *e : edges_)
    if (!e->src_ && e->dst_) pending[e->dst_->id_] -= 1;

  std::deque<Node *> ready;
  for (size_t i = 0; i < nodes_.size(); ++i)
    if (pending[i] == 0) {
      begin_nodes_.push_back(nodes_[i]);
      ready.push_back(nodes_[i]);
    }
  while (!ready.empty()) {
    Node *cur = ready.front();
    ready.pop_front();
    plan_.push_back(cur->id_);
    for (Edge *e : cur->out_) {
      Node *nxt = e->dst_;
      if (!nxt) continue;
      if (--pending[nxt->id_] <= 0) {
        ready.push_back(nxt);
        next_nodes_[cur->id_].push_back(nxt);
      }
    }
    MarkUse(cur, blocks_, touched);
    for (auto *info : touched) info->used_.push_back(cur);
  }
}

void Graph::AnalyzeEdges() {
  // A block may be released after its last user when it is produced inside
  // the graph and nobody outside the graph (i.e. no live tensor handle beyond
  // the copies captured by the closures) still refers to it.
  for (auto *info : infos_) {
    if (info->used_.empty()) continue;
    if ((info->type_ == BlockType::kInter || info->type_ == BlockType::kEnd) &&
        info->graph_ref_ >= info->blk_->ref_count()) {
      free_blocks_[info->used_.back()->id_].push_back(info->blk_);
    }
  }
}

void Graph::ExecNode(Node *n) {
  if (profiling_now() && hooks_.timed_exec && n->name_ != "Waiting")
    hooks_.timed_exec(*n);
  else
    n->op_();
}

size_t Graph::Prepare(bool serial) {
  if (serial != in_serial_) {
    in_serial_ = serial;
    dirty_ = true;
  }
  if (dirty_) Analyze();
  return plan_.size();
}

void Graph::RunRange(size_t first, size_t last) {
  last = std::min(last, plan_.size());
  for (size_t s = first; s < last; ++s) {
    int id = plan_[s];
    ExecNode(nodes_[id]);
    for (Block *b : free_blocks_[id]) b->free_data();
  }
}

void Graph::RunPlan() {
  Clock::time_point start;
  const bool prof = profiling_now();
  if (prof) {
    if (hooks_.sync) hooks_.sync();
    start = Clock::now();
  }
  RunRange(0, plan_.size());
  ++iteration_;
  if (prof) {
    if (hooks_.sync) hooks_.sync();
    time_elapsed_ += std::chrono::duration<float>(Clock::now() - start).count();
    if (hooks_.collect)
      for (Node *n : nodes_)
        if (n->name_ != "Waiting") n->elapsed_ += hooks_.collect(*n);
  }
}

void Graph::RunGraph() {
  Prepare(false);
  RunPlan();
}

void Graph::RunInSerial() {
  Prepare(true);
  RunPlan();
}

static const char *TypeName(BlockType t) {
  switch (t) {
    case BlockType::kInput: return "input";
    case BlockType::kParam: return "param";
    case BlockType::kInter: return "inter";
    case BlockType::kEnd: return "end";
    default: return "unknown";
  }
}

std::string Graph::Debug() const {
  std::ostringstream ss;
  ss << "graph: " << nodes_.size() << " nodes, " << edges_.size() << " edges, "
     << infos_.size() << " blocks\n";
  for (Node *n : nodes_) {
    ss << "  node " << n->id_ << " [" << n->name_ << "] in:";
    for (Edge *e : n->in_)
      ss << ' ' << (e->src_ ? std::to_string(e->src_->id_) : "-") << "/b"
         << blocks_.at(e->blk_)->id_;
    ss << " out:";
    for (Edge *e : n->out_)
      ss << ' ' << (e->dst_ ? std::to_string(e->dst_->id_) : "-") << "/b"
         << blocks_.at(e->blk_)->id_;
    ss << '\n';
  }
  for (auto *info : infos_) {
    ss << "  block " << info->id_ << " type=" << TypeName(info->type_)
       << " graph_ref=" << info->graph_ref_
       << " ref=" << info->blk_->ref_count() << " users:";
    for (Node *n : info->used_) ss << ' ' << n->id_;
    ss << '\n';
  }
  if (!dirty_) {
    ss << "  plan:";
    for (int id : plan_) ss << ' ' << id;
    ss << '\n';
    for (size_t i = 0; i < free_blocks_.size(); ++i) {
      if (free_blocks_[i].empty()) continue;
      ss << "  after node " << i << " free:";
      for (Block *b : free_blocks_[i]) ss << " b" << blocks_.at(b)->id_;
      ss << '\n';
    }
  }
  return ss.str();
}

std::string Graph::TimeProfiling() const {
  std::ostringstream ss;
  const int measured = iteration_ - skip_iteration_;
  if (verbosity_ <= 0 || measured <= 0) return ss.str();
  ss << "\nTime Profiling:\n";
  if (verbosity_ == 1) {
    // everything before the first op whose name contains "Backward" counts
    // as forward propagation
    bool forward = true;
    float fwd = 0.f;
    for (Node *n : nodes_) {
      if (n->elapsed_ <= 0) continue;
      if (forward && n->name_.find("Backward") != std::string::npos)
        forward = false;
      if (forward) fwd += n->elapsed_ / measured;
    }
    float total = time_elapsed_ / measured;
    ss << "Forward Propagation Time : " << fwd << " sec\n";
    ss << "Backward Propagation Time : " << (total - fwd) << " sec\n";
  } else {
    for (Node *n : nodes_) {
      if (n->elapsed_ <= 0) continue;
      if (verbosity_ == 3 && n->name_.find("Dist") == std::string::npos)
        continue;
      ss << "OP_ID" << n->id_ << ". " << n->name_ << " : "
         << n->elapsed_ / measured << " sec\n";
    }
  }
  return ss.str();
}

}  // namespace singa_amd

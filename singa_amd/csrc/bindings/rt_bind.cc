// pybind11 surface of the native runtime: Block handles, the graph scheduler
// and the native IO / utility classes.  No HIP symbols are referenced here so
// the module also loads on hosts without a GPU.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <atomic>

#include "../runtime/binfile.h"
#include "../runtime/block.h"
#include "../runtime/channel.h"
#include "../runtime/graph.h"
#include "../runtime/lmdbfile.h"
#include "../runtime/logging.h"
#include "../runtime/snapshot.h"
#include "../runtime/textfile.h"
#include "../runtime/timer.h"

namespace py = pybind11;
using namespace singa_amd;

namespace {

std::atomic<int64_t> g_block_ids{0};

// Block whose storage is an arbitrary Python object (a flat torch tensor taken
// from the caching allocator of the owning device).
class PyBlock : public Block {
 public:
  explicit PyBlock(size_t size) : Block(size, g_block_ids++), data_(py::none()) {}
  bool initialized() const override { return !data_.is_none(); }
  void free_data() override { data_ = py::none(); }
  py::object get() const { return data_; }
  void set(py::object o) { data_ = std::move(o); }

 private:
  py::object data_;
};

BlockVec ToBlocks(const std::vector<std::shared_ptr<PyBlock>> &v) {
  BlockVec out;
  out.reserve(v.size());
  for (auto &b : v) out.push_back(b.get());
  return out;
}

// Graph that keeps the Python block objects alive for as long as they are
// part of the recorded graph.
class PyGraph : public Graph {
 public:
  void Add(py::function fn, const std::vector<std::shared_ptr<PyBlock>> &reads,
           const std::vector<std::shared_ptr<PyBlock>> &writes,
           const std::string &name) {
    for (auto &b : reads) keep_.push_back(b);
    for (auto &b : writes) keep_.push_back(b);
    AddOperation([fn]() { fn(); }, ToBlocks(reads), ToBlocks(writes), name);
  }
  void ResetAll() {
    Reset();
    keep_.clear();
  }
  std::shared_ptr<PyBlock> Share(Block *b) const {
    for (auto &k : keep_)
      if (k.get() == b) return k;
    return nullptr;
  }

 private:
  std::vector<std::shared_ptr<PyBlock>> keep_;
};

}  // namespace

PYBIND11_MODULE(_rt, m) {
  m.doc() = "singa_amd native runtime (scheduler, blocks, IO, utils)";

  py::class_<PyBlock, std::shared_ptr<PyBlock>>(m, "Block")
      .def(py::init<size_t>(), py::arg("size"))
      .def_property("data", &PyBlock::get, &PyBlock::set)
      .def("size", &PyBlock::size)
      .def("id", &PyBlock::id)
      .def("initialized", &PyBlock::initialized)
      .def("free_data", &PyBlock::free_data)
      .def("IncRefCount", &PyBlock::IncRefCount)
      .def("DecRefCount", &PyBlock::DecRefCount)
      .def("ref_count", &PyBlock::ref_count);

  py::enum_<BlockType>(m, "BlockType")
      .value("kUnknown", BlockType::kUnknown)
      .value("kInput", BlockType::kInput)
      .value("kParam", BlockType::kParam)
      .value("kInter", BlockType::kInter)
      .value("kEnd", BlockType::kEnd);

  py::class_<PyGraph>(m, "Graph")
      .def(py::init<>())
      .def("AddOperation", &PyGraph::Add, py::arg("fn"), py::arg("reads"),
           py::arg("writes"), py::arg("name") = "no_name")
      .def("Reset", &PyGraph::ResetAll)
      .def("RunGraph", &PyGraph::RunGraph)
      .def("RunInSerial", &PyGraph::RunInSerial)
      .def("Prepare", &PyGraph::Prepare)
      .def("RunRange", &PyGraph::RunRange)
      .def("FinishIteration", &PyGraph::FinishIteration)
      .def("set_force_profile", &PyGraph::set_force_profile)
      .def("AddNodeTime", &PyGraph::AddNodeTime)
      .def("AddTimeElapsed", &PyGraph::AddTimeElapsed)
      .def("Debug", &PyGraph::Debug)
      .def("TimeProfiling", &PyGraph::TimeProfiling)
      .def("set_verbosity", &PyGraph::set_verbosity)
      .def("verbosity", &PyGraph::verbosity)
      .def("set_skip_iteration", &PyGraph::set_skip_iteration)
      .def("skip_iteration", &PyGraph::skip_iteration)
      .def("iteration", &PyGraph::iteration)
      .def("time_elapsed", &PyGraph::time_elapsed)
      .def("dirty", &PyGraph::dirty)
      .def("num_nodes", [](const PyGraph &g) { return g.nodes().size(); })
      .def("num_edges", [](const PyGraph &g) { return g.edges().size(); })
      .def("num_blocks", [](const PyGraph &g) { return g.block_infos().size(); })
      .def("plan", &PyGraph::plan)
      .def("run_node", [](PyGraph &g, int id) { g.nodes().at(id)->Run(); })
      .def("node_name",
           [](const PyGraph &g, int id) { return g.nodes().at(id)->op_name(); })
      .def("node_names",
           [](const PyGraph &g) {
             std::vector<std::string> out;
             for (auto *n : g.nodes()) out.push_back(n->op_name());
             return out;
           })
      .def("node_time",
           [](const PyGraph &g, int id) { return g.nodes().at(id)->time_elapsed(); })
      // (src, dst) node ids of the in / out edges of a node; -1 = none
      .def("node_in_edges",
           [](const PyGraph &g, int id) {
             std::vector<std::tuple<int, int, int64_t>> out;
             for (auto *e : g.nodes().at(id)->in_edges())
               out.emplace_back(e->src_node() ? e->src_node()->id() : -1,
                                e->dst_node() ? e->dst_node()->id() : -1,
                                e->block()->id());
             return out;
           })
      .def("node_out_edges",
           [](const PyGraph &g, int id) {
             std::vector<std::tuple<int, int, int64_t>> out;
             for (auto *e : g.nodes().at(id)->out_edges())
               out.emplace_back(e->src_node() ? e->src_node()->id() : -1,
                                e->dst_node() ? e->dst_node()->id() : -1,
                                e->block()->id());
             return out;
           })
      .def("edge",
           [](const PyGraph &g, int id) {
             auto *e = g.edges().at(id);
             return std::make_tuple(e->src_node() ? e->src_node()->id() : -1,
                                    e->dst_node() ? e->dst_node()->id() : -1,
                                    e->block()->id());
           })
      .def("block_info",
           [](const PyGraph &g, std::shared_ptr<PyBlock> b) -> py::object {
             BlkInfo *info = g.block(b.get());
             if (!info) return py::none();
             py::dict d;
             d["id"] = info->id();
             d["type"] = info->type();
             d["graph_ref"] = info->graph_ref();
             auto *w = info->write_edge();
             d["write_edge"] =
                 w ? py::cast(std::make_tuple(
                         w->src_node() ? w->src_node()->id() : -1,
                         w->dst_node() ? w->dst_node()->id() : -1))
                   : py::object(py::none());
             std::vector<int> used;
             for (auto *n : info->used_nodes()) used.push_back(n->id());
             d["used_nodes"] = used;
             return d;
           })
      .def("leaf_block_ids",
           [](const PyGraph &g) {
             std::vector<int64_t> out;
             for (auto *b : g.leaf_blocks()) out.push_back(b->id());
             return out;
           })
      .def("begin_nodes",
           [](const PyGraph &g) {
             std::vector<int> out;
             for (auto *n : g.begin_nodes()) out.push_back(n->id());
             return out;
           })
      .def("next_nodes",
           [](const PyGraph &g) {
             std::vector<std::vector<int>> out;
             for (auto &v : g.next_nodes()) {
               out.emplace_back();
               for (auto *n : v) out.back().push_back(n->id());
             }
             return out;
           })
      .def("free_blocks",
           [](const PyGraph &g) {
             std::vector<std::vector<int64_t>> out;
             for (auto &v : g.free_blocks()) {
               out.emplace_back();
               for (auto *b : v) out.back().push_back(b->id());
             }
             return out;
           })
      .def("set_hooks",
           [](PyGraph &g, py::object timed_exec, py::object sync,
              py::object collect) {
             ExecHooks h;
             if (!timed_exec.is_none())
               h.timed_exec = [timed_exec](Node &n) {
                 timed_exec(n.id(), n.op_name());
               };
             if (!sync.is_none()) h.sync = [sync]() { sync(); };
             if (!collect.is_none())
               h.collect = [collect](Node &n) {
                 return collect(n.id()).cast<float>();
               };
             g.set_hooks(std::move(h));
           },
           py::arg("timed_exec") = py::none(), py::arg("sync") = py::none(),
           py::arg("collect") = py::none());

  // ---- logging ------------------------------------------------------------
  m.def("InitLogging", &InitLogging);
  m.def("LogToStderr", &LogToStderr);
  m.def("SetStderrLogging", &SetStderrLogging);
  m.def("SetLogDestination", &SetLogDestination);
  // severity 3 (FATAL) logs and then raises, like LOG(FATAL) in native code
  m.def("Log",
        [](int severity, const std::string &msg, const std::string &file,
           int line) {
          LogMessage(severity, msg, file, line);
          if (severity >= 3) throw std::runtime_error(msg);
        },
        py::arg("severity"), py::arg("msg"), py::arg("file") = "python",
        py::arg("line") = 0);

  // ---- channel ------------------------------------------------------------
  py::class_<Channel>(m, "Channel")
      .def("name", &Channel::name)
      .def("EnableDestStderr", &Channel::EnableDestStderr)
      .def("EnableDestFile", &Channel::EnableDestFile)
      .def("SetDestFilePath", &Channel::SetDestFilePath)
      .def("Send", &Channel::Send);
  m.def("InitChannel", &InitChannel, py::arg("argv0") = "");
  m.def("SetChannelDirectory", &SetChannelDirectory);
  m.def("GetChannel", &GetChannel, py::return_value_policy::reference);

  // ---- timer --------------------------------------------------------------
  py::class_<Timer>(m, "Timer")
      .def(py::init<>())
      .def("Tick", &Timer::Tick)
      .def("ElapsedMs", &Timer::ElapsedMs)
      .def("ElapsedSec", &Timer::ElapsedSec);

  // ---- record files ---------------------------------------------------------
  py::class_<BinFileWriter>(m, "BinFileWriter")
      .def(py::init<>())
      .def("Open", &BinFileWriter::Open, py::arg("path"),
           py::arg("append") = false, py::arg("capacity") = 10485760)
      .def("Close", &BinFileWriter::Close)
      .def("Write",
           [](BinFileWriter &w, const std::string &key, py::bytes val) {
             return w.Write(key, std::string(val));
           })
      .def("Flush", &BinFileWriter::Flush)
      .def("path", &BinFileWriter::path);
  py::class_<BinFileReader>(m, "BinFileReader")
      .def(py::init<>())
      .def("Open", &BinFileReader::Open, py::arg("path"),
           py::arg("capacity") = 10485760)
      .def("Close", &BinFileReader::Close)
      .def("Read",
           [](BinFileReader &r) -> py::object {
             std::string k, v;
             if (!r.Read(&k, &v)) return py::none();
             return py::make_tuple(k, py::bytes(v));
           })
      .def("Count", &BinFileReader::Count)
      .def("SeekToFirst", &BinFileReader::SeekToFirst)
      .def("path", &BinFileReader::path);
  py::class_<TextFileWriter>(m, "TextFileWriter")
      .def(py::init<>())
      .def("Open", &TextFileWriter::Open, py::arg("path"),
           py::arg("append") = false)
      .def("Close", &TextFileWriter::Close)
      .def("Write", &TextFileWriter::Write)
      .def("Flush", &TextFileWriter::Flush);
  py::class_<TextFileReader>(m, "TextFileReader")
      .def(py::init<>())
      .def("Open", &TextFileReader::Open)
      .def("Close", &TextFileReader::Close)
      .def("Read",
           [](TextFileReader &r) -> py::object {
             std::string k, v;
             if (!r.Read(&k, &v)) return py::none();
             return py::make_tuple(k, v);
           })
      .def("Count", &TextFileReader::Count)
      .def("SeekToFirst", &TextFileReader::SeekToFirst);

  py::class_<LMDBWriter>(m, "LMDBWriter")
      .def(py::init<>())
      .def("Open", &LMDBWriter::Open, py::arg("path"),
           py::arg("append") = false, py::arg("page_size") = 4096)
      .def("Close", &LMDBWriter::Close)
      .def("Write",
           [](LMDBWriter &w, py::bytes key, py::bytes val) {
             return w.Write(std::string(key), std::string(val));
           })
      .def("Flush", &LMDBWriter::Flush)
      .def("path", &LMDBWriter::path);
  py::class_<LMDBReader>(m, "LMDBReader")
      .def(py::init<>())
      .def("Open", &LMDBReader::Open)
      .def("Close", &LMDBReader::Close)
      .def("Read",
           [](LMDBReader &r) -> py::object {
             std::string k, v;
             if (!r.Read(&k, &v)) return py::none();
             return py::make_tuple(py::bytes(k), py::bytes(v));
           })
      .def("Count", &LMDBReader::Count)
      .def("SeekToFirst", &LMDBReader::SeekToFirst)
      .def("valid", &LMDBReader::valid)
      .def("depth", &LMDBReader::depth)
      .def("page_size", &LMDBReader::page_size)
      .def("path", &LMDBReader::path);

  // ---- snapshot -------------------------------------------------------------
  py::class_<SnapshotTensor>(m, "SnapshotTensor")
      .def(py::init<>())
      .def_readwrite("shape", &SnapshotTensor::shape)
      .def_readwrite("dtype", &SnapshotTensor::dtype)
      .def_property(
          "data",
          [](const SnapshotTensor &t) { return py::bytes(t.data); },
          [](SnapshotTensor &t, py::bytes b) { t.data = std::string(b); });
  py::class_<Snapshot>(m, "Snapshot")
      .def(py::init<const std::string &, bool, int>(), py::arg("prefix"),
           py::arg("write"), py::arg("max_param_size_mb") = 10)
      .def("Write", &Snapshot::Write)
      .def("Read", &Snapshot::Read)
      .def("ReadShape", &Snapshot::ReadShape)
      .def("Names", &Snapshot::Names)
      .def("Close", &Snapshot::Close);
  m.def("EncodeTensorProto", [](const SnapshotTensor &t) {
    return py::bytes(EncodeTensorProto(t));
  });
  m.def("DecodeTensorProto", [](py::bytes b) {
    return DecodeTensorProto(std::string(b));
  });
}

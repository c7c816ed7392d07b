// C++ host layer above the C ABI: the reference's Nim interface for the BN254 proof-input path, name for name.
//
// The reference's host code is Nim (reference/nim/proof_input/src); no Nim toolchain exists in the build image,
// so this mirror is C++ (the reference compiles to native code through C).  Every function keeps the name,
// argument meaning and failure behaviour of the proc it mirrors (Nim `assert` -> AssertionDefect here) and does
// its hashing through libcodex_p2.so, i.e. on the GPU.  nim/codex_p2.nim binds the same C ABI for Nim callers.
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/codex_p2.h"

namespace codex {

struct AssertionDefect : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline void doAssert(bool ok, const std::string& msg) {
  if (!ok) throw AssertionDefect(msg);
}

// ---- types.nim:6-109, types/bn254.nim:20-27 ---------------------------------------------------
using F = std::array<uint8_t, 32>;   // canonical little-endian field element
using Hash = F;
using Root = Hash;
using Entropy = F;
using Cell = std::vector<uint8_t>;
using Block = std::vector<uint8_t>;
using Seed = uint64_t;
using CellIdx = int64_t;
using BlockIdx = int64_t;
using SlotIdx = int64_t;

struct MerkleProof {            // types.nim:12-17
  int64_t leafIndex = 0;
  Hash leafValue{};
  std::vector<Hash> merklePath;
  int64_t numberOfLeaves = 0;
};
struct MerkleTree {             // types.nim:19-22: first layer = bottom, last = root
  std::vector<std::vector<Hash>> layers;
};

enum class DataSourceKind { SlotFile, FakeData };
struct DataSource {             // types.nim:65-74
  DataSourceKind kind = DataSourceKind::FakeData;
  std::string filename;
  Seed seed = 0;
};
struct SlotConfig { int64_t nCells = 0, nSamples = 0; DataSource dataSrc; };
struct DataSetConfig { int64_t nSlots = 0, nCells = 0, nSamples = 0; DataSource dataSrc; };
struct GlobalConfig { int64_t maxDepth = 0, maxLog2NSlots = 0, cellSize = 0, blockSize = 0; };
enum class FieldSelect { BN254, Goldilocks };
enum class HashSelect { Poseidon2, Monolith };
enum class FieldHashCombo { BN254_Poseidon2, Goldilocks_Poseidon2, Goldilocks_Monolith };
struct HashConfig { FieldSelect field = FieldSelect::BN254; HashSelect hashFun = HashSelect::Poseidon2; FieldHashCombo combo = FieldHashCombo::BN254_Poseidon2; };

struct CellProofInput { Cell cellData; MerkleProof merkleProof; };
struct SlotProofInput {         // types.nim:52-60
  Hash dataSetRoot{}, entropy{};
  int64_t nSlots = 0, nCells = 0;
  Hash slotRoot{};
  SlotIdx slotIndex = 0;
  MerkleProof slotProof;
  std::vector<CellProofInput> proofInputs;
};

inline int64_t cellsPerBlock(const GlobalConfig& glob) {   // types.nim:104-107
  doAssert(glob.cellSize > 0 && glob.blockSize % glob.cellSize == 0, "block size is not divisible by cell size");
  return glob.blockSize / glob.cellSize;
}

// ---- misc.nim:10-35 -------------------------------------------------------------------------------
inline int floorLog2(int64_t x) { int k = -1; while (x > 0) { ++k; x >>= 1; } return k; }
inline int ceilingLog2(int64_t x) { return x == 0 ? -1 : floorLog2(x - 1) + 1; }
inline int exactLog2(int64_t x) { int k = ceilingLog2(x); doAssert(k >= 0 && x == (int64_t(1) << k), "exactLog2: not a power of two"); return k; }
inline int64_t checkPowerOfTwo(int64_t x, const std::string& what) {
  int k = ceilingLog2(x);
  doAssert(k >= 0 && x == (int64_t(1) << k), "`" + what + "` is expected to be a power of 2");
  return x;
}
inline int64_t pow2(int k) { return int64_t(1) << k; }

// ---- the engine -----------------------------------------------------------------------------------
// Every GPU of the node behind one object (cp2_multi, include/codex_p2.h section e): the seam calls (hashCell, compress,
// merkleTree, cellIndices ...) run on the first device's context, generateProofInputBN254 cuts the dataset's slots over all
// devices.  Engine() takes ONE device -- the first visible gfx950 -- unless the environment variable CODEX_P2_GPUS names more ("all",
// "<count>" or an index list): several devices are opt-in until the exchange between two real devices has a committed record
// (include/codex_p2.h, cp2_multi_init); Engine(d) exactly device d.
class Engine {
 public:
  Engine() { init(nullptr, 0); }
  explicit Engine(int device) { init(&device, 1); }
  explicit Engine(const std::vector<int>& devices) { init(devices.data(), (int)devices.size()); }
  ~Engine() { cp2_multi_free(multi_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  cp2_multi* multi() const { return multi_; }
  cp2_ctx* ctx() const {
    cp2_ctx* c = cp2_multi_ctx(multi_, 0);
    if (!c) throw std::runtime_error(std::string("cp2_init: ") + cp2_strerror(CP2_ERR_NO_DEVICE));
    return c;
  }
  void check(int st, const char* what) const {
    if (st == CP2_OK) return;
    std::string msg = std::string(what) + ": " + cp2_strerror(st);
    const char* d = cp2_multi_last_error(multi_);
    if (!d || !*d) d = cp2_last_error(cp2_multi_ctx(multi_, 0));
    if (d && *d) msg += std::string(" (") + d + ")";
    if (st == CP2_ERR_INVALID) throw AssertionDefect(msg);
    throw std::runtime_error(msg);
  }
 private:
  // The header this program was COMPILED against and the library it found at run time must be of one major version, the
  // library's minor at least the header's (include/codex_p2.h, CP2_ABI_VERSION_*): refused with both numbers otherwise.
  static void requireAbi() {
    const int v = cp2_abi_version();
    if ((v >> 16) != CP2_ABI_VERSION_MAJOR || (v & 0xffff) < CP2_ABI_VERSION_MINOR)
      throw std::runtime_error("libcodex_p2.so has ABI version " + std::to_string(v >> 16) + "." + std::to_string(v & 0xffff) + ", this program was built against " +
                               std::to_string(CP2_ABI_VERSION_MAJOR) + "." + std::to_string(CP2_ABI_VERSION_MINOR));
  }
  void init(const int* devices, int n) {
    requireAbi();
    int st = cp2_multi_init(devices, n, &multi_);
    if (st != CP2_OK) {
      char why[512] = "";
      if (st == CP2_ERR_INVALID) (void)cp2_check_environment(why, sizeof why);   // names the CODEX_P2_* variable that does not hold what it takes
      throw std::runtime_error(std::string("cp2_multi_init: ") + cp2_strerror(st) + (why[0] ? std::string(" (") + why + ")" : std::string()));
    }
  }
  cp2_multi* multi_ = nullptr;
};

// ---- types/bn254.nim:27-59 --------------------------------------------------------------------------
inline F toF(uint64_t x) { F f{}; std::memcpy(f.data(), &x, 8); return f; }
inline F intToBN254(int64_t x) { doAssert(x >= 0, "negative integers are not supported"); return toF((uint64_t)x); }
inline uint64_t extractLowBits(const F& fld, int k) {   // types/bn254.nim:47-59
  doAssert(k > 0 && k <= 64, "extractLowBits: k out of range");
  uint64_t lo; std::memcpy(&lo, fld.data(), 8);
  return k == 64 ? lo : (lo & ((uint64_t(1) << k) - 1));
}

// ---- merkle/bn254.nim:18-63 -----------------------------------------------------------------------
inline F compressWithKey(Engine& e, int key, const F& x, const F& y) {
  uint8_t xy[64]; std::memcpy(xy, x.data(), 32); std::memcpy(xy + 32, y.data(), 32);
  F out{};
  e.check(cp2_compress_batch(e.ctx(), xy, (uint32_t)key, out.data(), 1), "compressWithKey");
  return out;
}
inline MerkleTree merkleTreeBN254(Engine& e, const std::vector<F>& xs) {
  doAssert(!xs.empty(), "merkleTree: input is empty");
  size_t total = cp2_merkle_total(xs.size()), nl = 0;
  std::vector<uint8_t> flat(total * 32);
  std::vector<size_t> sizes(cp2_merkle_num_layers(xs.size()));
  e.check(cp2_merkle_tree(e.ctx(), xs[0].data(), xs.size(), flat.data(), sizes.data(), &nl), "merkleTreeBN254");
  MerkleTree t;
  size_t off = 0;
  for (size_t k = 0; k < nl; ++k) {
    std::vector<Hash> layer(sizes[k]);
    std::memcpy(layer[0].data(), &flat[off * 32], sizes[k] * 32);
    off += sizes[k];
    t.layers.push_back(std::move(layer));
  }
  return t;
}
inline F merkleDigestBN254(Engine& e, const std::vector<F>& xs) {
  doAssert(!xs.empty(), "Merkle.digest: input is empty");
  F out{};
  e.check(cp2_merkle_root(e.ctx(), xs[0].data(), xs.size(), out.data()), "merkleDigestBN254");
  return out;
}

// ---- merkle.nim:6-100 -------------------------------------------------------------------------------
inline int treeDepth(const MerkleTree& t) { return (int)t.layers.size() - 1; }
inline int64_t treeNumberOfLeaves(const MerkleTree& t) { return (int64_t)t.layers[0].size(); }
inline Hash treeRoot(const MerkleTree& t) { doAssert(t.layers.back().size() == 1, "treeRoot"); return t.layers.back()[0]; }
inline MerkleProof merkleProof(const MerkleTree& tree, int64_t index) {
  int depth = treeDepth(tree);
  int64_t nleaves = treeNumberOfLeaves(tree);
  doAssert(index >= 0 && index < nleaves, "merkleProof: index out of range");
  MerkleProof p;
  p.merklePath.resize(depth);
  int64_t k = index, m = nleaves;
  for (int i = 0; i < depth; ++i) {
    int64_t j = k ^ 1;
    p.merklePath[i] = (j < m) ? tree.layers[i][j] : Hash{};
    k >>= 1;
    m = (m + 1) >> 1;
  }
  p.leafIndex = index; p.leafValue = tree.layers[0][index]; p.numberOfLeaves = nleaves;
  return p;
}
inline Hash reconstructRoot(Engine& e, const MerkleProof& proof) {
  int64_t m = proof.numberOfLeaves, j = proof.leafIndex;
  Hash h = proof.leafValue;
  int bottomFlag = 1;
  for (const Hash& p : proof.merklePath) {
    if (j & 1) h = compressWithKey(e, bottomFlag, p, h);
    else if (j == m - 1) h = compressWithKey(e, bottomFlag + 2, h, p);
    else h = compressWithKey(e, bottomFlag, h, p);
    bottomFlag = 0;
    j >>= 1;
    m = (m + 1) >> 1;
  }
  return h;
}
inline bool checkMerkleProof(Engine& e, const Hash& root, const MerkleProof& proof) { return root == reconstructRoot(e, proof); }
inline MerkleProof mergeMerkleProofs(Engine& e, const MerkleProof& bottomProof, const MerkleProof& topProof) {
  doAssert(reconstructRoot(e, bottomProof) == topProof.leafValue, "mergeMerkleProofs: bottom root != top leaf");
  MerkleProof p;
  p.leafIndex = topProof.leafIndex * bottomProof.numberOfLeaves + bottomProof.leafIndex;
  p.leafValue = bottomProof.leafValue;
  p.numberOfLeaves = bottomProof.numberOfLeaves * topProof.numberOfLeaves;
  p.merklePath = bottomProof.merklePath;
  p.merklePath.insert(p.merklePath.end(), topProof.merklePath.begin(), topProof.merklePath.end());
  return p;
}
inline MerkleProof padMerkleProof(const MerkleProof& old, int64_t newlen) {   // types.nim:27-37
  doAssert((int64_t)old.merklePath.size() <= newlen, "padMerkleProof: pad >= 0");
  MerkleProof p = old;
  p.merklePath.resize(newlen, Hash{});
  return p;
}

// ---- blocks/bn254.nim:17-67 -------------------------------------------------------------------------
inline MerkleTree merkleTree(Engine& e, const HashConfig& hashcfg, const std::vector<Hash>& what) {
  doAssert(hashcfg.combo == FieldHashCombo::BN254_Poseidon2, "merkleTree: BN254_Poseidon2 only");
  return merkleTreeBN254(e, what);
}
inline Hash hashCell(Engine& e, const HashConfig& hashcfg, const GlobalConfig& globcfg, const Cell& cellData) {
  doAssert(hashcfg.field == FieldSelect::BN254 && hashcfg.hashFun == HashSelect::Poseidon2, "hashCell: BN254/Poseidon2 only");
  doAssert((int64_t)cellData.size() == globcfg.cellSize, "cells are expected to be exactly " + std::to_string(globcfg.cellSize) + " bytes");
  Hash out{};
  e.check(cp2_hash_cells(e.ctx(), cellData.data(), cellData.size(), 1, out.data()), "hashCell");
  return out;
}
inline std::vector<Hash> hashCellsOfBlock(Engine& e, const GlobalConfig& globcfg, const Block& blockData) {
  doAssert((int64_t)blockData.size() == globcfg.blockSize, "network blocks are expected to be exactly" + std::to_string(globcfg.blockSize) + " bytes");
  std::vector<Hash> leaves(cellsPerBlock(globcfg));
  e.check(cp2_hash_cells(e.ctx(), blockData.data(), globcfg.cellSize, leaves.size(), leaves[0].data()), "hashCell");
  return leaves;
}
inline Hash hashNetworkBlock(Engine& e, const HashConfig&, const GlobalConfig& globcfg, const Block& blockData) {
  return merkleDigestBN254(e, hashCellsOfBlock(e, globcfg, blockData));
}
inline MerkleTree networkBlockTree(Engine& e, const HashConfig& hashcfg, const GlobalConfig& globcfg, const Block& blockData) {
  doAssert(hashcfg.field == FieldSelect::BN254, "networkBlockTree: BN254 only");
  return merkleTree(e, hashcfg, hashCellsOfBlock(e, globcfg, blockData));
}

// ---- sample/bn254.nim:16-27 -------------------------------------------------------------------------
inline std::vector<int64_t> cellIndices(Engine& e, const HashConfig& hashcfg, const Entropy& entropy, const Root& slotRoot,
                                        int64_t numberOfCells, int64_t nSamples) {
  doAssert(hashcfg.field == FieldSelect::BN254, "cellIndex: BN254 only");
  int log2 = ceilingLog2(numberOfCells);
  doAssert(log2 >= 0 && (int64_t(1) << log2) == numberOfCells, "for this version, `numberOfCells` is assumed to be a power of two");
  std::vector<uint64_t> raw(nSamples);
  e.check(cp2_cell_indices(e.ctx(), entropy.data(), slotRoot.data(), (uint64_t)numberOfCells, (size_t)nSamples, raw.data()), "cellIndices");
  return std::vector<int64_t>(raw.begin(), raw.end());
}
inline int64_t cellIndex(Engine& e, const HashConfig& hashcfg, const Entropy& entropy, const Root& slotRoot, int64_t numberOfCells, int64_t counter) {
  doAssert(counter >= 1, "cellIndex: counters start at 1");
  return cellIndices(e, hashcfg, entropy, slotRoot, numberOfCells, counter).back();
}

// ---- slot.nim:23-73, dataset.nim:32-51 ----------------------------------------------------------------
inline Seed parametricSlotSeed(Seed seed, SlotIdx k) { return cp2_slot_seed(seed, (uint64_t)k); }
inline std::string parametricSlotFileName(const std::string& basefile, SlotIdx k) { return basefile + std::to_string(k) + ".dat"; }
inline SlotConfig slotCfgFromDataSetCfg(const DataSetConfig& d, SlotIdx idx) {
  doAssert(idx >= 0 && idx < d.nSlots, "slotCfgFromDataSetCfg: slot index out of range");
  SlotConfig s; s.nCells = d.nCells; s.nSamples = d.nSamples; s.dataSrc = d.dataSrc;
  if (d.dataSrc.kind == DataSourceKind::FakeData) s.dataSrc.seed = parametricSlotSeed(d.dataSrc.seed, idx);
  else s.dataSrc.filename = parametricSlotFileName(d.dataSrc.filename, idx);
  return s;
}
inline Cell slotLoadCellData(Engine& e, const GlobalConfig& globcfg, const SlotConfig& cfg, CellIdx idx) {
  Cell cell(globcfg.cellSize, 0);
  if (cfg.dataSrc.kind == DataSourceKind::FakeData) {
    e.check(cp2_gen_fake_cells(e.ctx(), cfg.dataSrc.seed, (uint64_t)idx, 1, cell.size(), cell.data()), "genFakeCell");
  } else {
    FILE* f = std::fopen(cfg.dataSrc.filename.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + cfg.dataSrc.filename);
    if (std::fseek(f, (long)(globcfg.cellSize * idx), SEEK_SET) == 0) (void)!std::fread(cell.data(), 1, cell.size(), f);
    std::fclose(f);
  }
  return cell;
}
inline Block slotLoadBlockData(Engine& e, const GlobalConfig& globcfg, const SlotConfig& cfg, BlockIdx idx) {
  Block b;
  for (int64_t i = 0; i < cellsPerBlock(globcfg); ++i) {
    Cell c = slotLoadCellData(e, globcfg, cfg, idx * cellsPerBlock(globcfg) + i);
    b.insert(b.end(), c.begin(), c.end());
  }
  return b;
}

// ---- gen_input/bn254.nim:35-79, json/bn254.nim:57-78 ------------------------------------------------
inline cp2_config toEngineConfig(const GlobalConfig& g, const DataSetConfig& d) {
  cp2_config c{};
  c.max_depth = (int32_t)g.maxDepth; c.max_log2_nslots = (int32_t)g.maxLog2NSlots;
  c.cell_size = (uint64_t)g.cellSize; c.block_size = (uint64_t)g.blockSize;
  c.n_slots = (uint64_t)d.nSlots; c.n_cells = (uint64_t)d.nCells; c.n_samples = (uint64_t)d.nSamples;
  c.seed = d.dataSrc.seed;
  c.file_base = d.dataSrc.kind == DataSourceKind::SlotFile ? d.dataSrc.filename.c_str() : nullptr;
  return c;
}

inline SlotProofInput generateProofInputBN254(Engine& e, const HashConfig& hashCfg, const GlobalConfig& globCfg,
                                              const DataSetConfig& dsetCfg, SlotIdx slotIdx, const Entropy& entropy) {
  doAssert(hashCfg.field == FieldSelect::BN254, "generateProofInputBN254: BN254 only");
  doAssert(dsetCfg.nCells % cellsPerBlock(globCfg) == 0, "nblocks * cellsPerBlock == ncells");
  doAssert(slotIdx >= 0 && slotIdx < dsetCfg.nSlots, "slot index out of range");
  cp2_config cfg = toEngineConfig(globCfg, dsetCfg);
  // all slot trees (gen_input/bn254.nim:41-42), the slots cut over the engine's devices; one gather of the slot roots and
  // the dataset tree (:49-51) on every device; the proof input comes from the device that holds slotIdx
  cp2_multi_dataset* ds = nullptr;
  // optional tree cache (not in the reference, which recomputes every tree on every run): CODEX_P2_CACHE=<file>
  const char* cache = std::getenv("CODEX_P2_CACHE");
  if (cache && *cache) e.check(cp2_multi_dataset_build_cached(e.multi(), &cfg, cache, &ds), "buildSlotTree (cached)");
  else e.check(cp2_multi_dataset_build(e.multi(), &cfg, &ds), "buildSlotTree (all slots)");
  std::shared_ptr<cp2_multi_dataset> ds_guard(ds, cp2_multi_dataset_free);
  cp2_proof_input* pi = nullptr;
  e.check(cp2_multi_proof_input_generate(ds, (uint64_t)slotIdx, entropy.data(), &pi), "generateProofInput");
  SlotProofInput out;
  std::shared_ptr<cp2_proof_input> pi_guard(pi, cp2_proof_input_free);   // `out` is a plain value, as in the reference
  cp2_proof_input_roots(pi, out.dataSetRoot.data(), out.slotRoot.data(), out.entropy.data());
  out.nSlots = dsetCfg.nSlots; out.nCells = dsetCfg.nCells; out.slotIndex = slotIdx;
  out.slotProof.leafIndex = slotIdx; out.slotProof.leafValue = out.slotRoot; out.slotProof.numberOfLeaves = dsetCfg.nSlots;
  out.slotProof.merklePath.resize(globCfg.maxLog2NSlots);
  if (globCfg.maxLog2NSlots) std::memcpy(out.slotProof.merklePath[0].data(), cp2_proof_input_slot_proof(pi), globCfg.maxLog2NSlots * 32);
  size_t ns = cp2_proof_input_nsamples(pi);
  const uint64_t* idx = cp2_proof_input_cell_indices(pi);
  const uint8_t* cells = cp2_proof_input_cell_data(pi);
  const uint8_t* paths = cp2_proof_input_merkle_paths(pi);
  const uint8_t* leaves = cp2_proof_input_leaf_hashes(pi);
  for (size_t i = 0; i < ns; ++i) {
    CellProofInput c;
    c.cellData.assign(cells + i * globCfg.cellSize, cells + (i + 1) * globCfg.cellSize);
    c.merkleProof.leafIndex = (int64_t)idx[i];
    if (leaves) std::memcpy(c.merkleProof.leafValue.data(), leaves + 32 * i, 32);   // mergeMerkleProofs keeps the bottom leaf, merkle.nim:94
    c.merkleProof.numberOfLeaves = dsetCfg.nCells;
    c.merkleProof.merklePath.resize(globCfg.maxDepth);
    if (globCfg.maxDepth) std::memcpy(c.merkleProof.merklePath[0].data(), paths + i * globCfg.maxDepth * 32, globCfg.maxDepth * 32);
    out.proofInputs.push_back(std::move(c));
  }
  return out;
}

// json/bn254.nim:77.  Works on ANY SlotProofInput value (as the reference's does): the fields are marshalled into an
// engine object (cp2_proof_input_create) and written by the byte-exact writer.
inline void exportProofInputBN254(const HashConfig& hashcfg, const std::string& fname, const SlotProofInput& prfInput) {
  doAssert(hashcfg.field == FieldSelect::BN254, "exportProofInputBN254: BN254 only");
  const size_t ns = prfInput.proofInputs.size();
  cp2_config cfg{};
  cfg.max_log2_nslots = (int32_t)prfInput.slotProof.merklePath.size();
  cfg.max_depth = ns ? (int32_t)prfInput.proofInputs[0].merkleProof.merklePath.size() : 0;
  cfg.cell_size = ns ? prfInput.proofInputs[0].cellData.size() : 0;
  cfg.n_slots = (uint64_t)prfInput.nSlots;
  cfg.n_cells = (uint64_t)prfInput.nCells;
  std::vector<uint8_t> proof((size_t)cfg.max_log2_nslots * 32), cells(ns * cfg.cell_size), paths(ns * (size_t)cfg.max_depth * 32);
  std::vector<uint64_t> idx(ns);
  for (size_t i = 0; i < proof.size() / 32; ++i) std::memcpy(&proof[32 * i], prfInput.slotProof.merklePath[i].data(), 32);
  for (size_t i = 0; i < ns; ++i) {
    const CellProofInput& c = prfInput.proofInputs[i];
    doAssert(c.cellData.size() == cfg.cell_size && c.merkleProof.merklePath.size() == (size_t)cfg.max_depth, "exportProofInput: ragged proof inputs");
    idx[i] = (uint64_t)c.merkleProof.leafIndex;
    std::memcpy(&cells[i * cfg.cell_size], c.cellData.data(), cfg.cell_size);
    for (size_t d = 0; d < (size_t)cfg.max_depth; ++d) std::memcpy(&paths[(i * cfg.max_depth + d) * 32], c.merkleProof.merklePath[d].data(), 32);
  }
  cp2_proof_input* pi = nullptr;
  int st = cp2_proof_input_create(&cfg, (uint64_t)prfInput.slotIndex, prfInput.dataSetRoot.data(), prfInput.entropy.data(), prfInput.slotRoot.data(),
                                  proof.data(), ns, idx.data(), cells.data(), paths.data(), nullptr, &pi);
  if (st == CP2_OK) {
    st = cp2_proof_input_write_json(pi, fname.c_str());
    cp2_proof_input_free(pi);
  }
  if (st != CP2_OK) throw std::runtime_error(std::string("exportProofInput: ") + cp2_strerror(st));
}

}  // namespace codex

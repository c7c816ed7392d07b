// Self-test of the C++ mirror of the Nim interface, modelled on the reference's own (stale) smoke test
// reference/nim/proof_input/src/testmain.nim:22-45 and reference/haskell/src/Poseidon2/Merkle.hs:136-152:
// for n = 1..N leaves toF(100+i), every extracted Merkle proof must reconstruct the root and
// Merkle.digest must equal treeRoot.  Prints one line per n and the roots in hex (checked by pytest against
// the oracle).  Also runs hashCell / networkBlockTree / mergeMerkleProofs / cellIndices once.
#include <cstdio>
#include <cstdlib>

#include "proof_input_api.hpp"

using namespace codex;

static void printHex(const char* label, const F& f) {
  std::printf("%s 0x", label);
  for (int i = 31; i >= 0; --i) std::printf("%02x", f[i]);
  std::printf("\n");
}

int main(int argc, char** argv) {
  int N = argc > 1 ? std::atoi(argv[1]) : 12;
  try {
    Engine e(0);
    HashConfig hashcfg;
    bool all = true;
    for (int n = 1; n <= N; ++n) {
      std::vector<F> xs;
      for (int i = 0; i < n; ++i) xs.push_back(toF(100 + i));
      MerkleTree tree = merkleTreeBN254(e, xs);
      Hash root = treeRoot(tree);
      bool ok = (merkleDigestBN254(e, xs) == root);
      for (int j = 0; j < n; ++j) ok = ok && checkMerkleProof(e, root, merkleProof(tree, j));
      std::printf("testing Merkle proofs for a tree with %d leaves: %s\n", n, ok ? "OK." : "FAILED!!");
      char label[32];
      std::snprintf(label, sizeof label, "root[%d]", n);
      printHex(label, root);
      all = all && ok;
    }
    // one network block of 8 cells x 64 bytes of fake data: hashCell, networkBlockTree, merged proof
    GlobalConfig glob{8, 4, 64, 512};
    SlotConfig scfg;
    scfg.nCells = 16;
    scfg.dataSrc.kind = DataSourceKind::FakeData;
    scfg.dataSrc.seed = parametricSlotSeed(12345, 2);
    Block b0 = slotLoadBlockData(e, glob, scfg, 0), b1 = slotLoadBlockData(e, glob, scfg, 1);
    MerkleTree t0 = networkBlockTree(e, hashcfg, glob, b0), t1 = networkBlockTree(e, hashcfg, glob, b1);
    doAssert(hashNetworkBlock(e, hashcfg, glob, b0) == treeRoot(t0), "hashNetworkBlock == root of networkBlockTree");
    Cell c3 = slotLoadCellData(e, glob, scfg, 3);
    doAssert(hashCell(e, hashcfg, glob, c3) == t0.layers[0][3], "hashCell == leaf 3");
    MerkleTree big = merkleTree(e, hashcfg, {treeRoot(t0), treeRoot(t1)});
    MerkleProof merged = padMerkleProof(mergeMerkleProofs(e, merkleProof(t1, 5), merkleProof(big, 1)), glob.maxDepth);
    printHex("slotRoot", treeRoot(big));
    printHex("cellHash[13]", merged.leafValue);
    std::printf("merged leafIndex %lld numberOfLeaves %lld pathLen %zu\n", (long long)merged.leafIndex,
                (long long)merged.numberOfLeaves, merged.merklePath.size());
    for (size_t i = 0; i < merged.merklePath.size(); ++i) {
      char label[32];
      std::snprintf(label, sizeof label, "path[%zu]", i);
      printHex(label, merged.merklePath[i]);
    }
    std::vector<int64_t> idx = cellIndices(e, hashcfg, intToBN254(1234567), treeRoot(big), 16, 6);
    std::printf("cellIndices");
    for (auto v : idx) std::printf(" %lld", (long long)v);
    std::printf("\n");
    // the two top-level procs with the reference's own signatures (gen_input/bn254.nim:78, json/bn254.nim:77):
    // every proof input carries the hash of its cell as leafValue, and a SlotProofInput VALUE (not an engine handle)
    // exports to the byte-exact text (argv[2] = directory: pi.json, and pi_edited.json with entropy replaced)
    {
      GlobalConfig g2{10, 3, 128, 1024};
      DataSetConfig d2;
      d2.nSlots = 5; d2.nCells = 64; d2.nSamples = 6;
      d2.dataSrc.kind = DataSourceKind::FakeData;
      d2.dataSrc.seed = 777;
      SlotProofInput pi = generateProofInputBN254(e, hashcfg, g2, d2, 3, intToBN254(31337));
      bool leaves_ok = pi.proofInputs.size() == 6;
      for (const CellProofInput& c : pi.proofInputs)
        leaves_ok = leaves_ok && hashCell(e, hashcfg, g2, c.cellData) == c.merkleProof.leafValue && c.merkleProof.numberOfLeaves == 64;
      std::printf("proof input leafValues: %s\n", leaves_ok ? "OK." : "FAILED!!");
      all = all && leaves_ok;
      if (argc > 2) {
        std::string dir = argv[2];
        exportProofInputBN254(hashcfg, dir + "/pi.json", pi);
        SlotProofInput edited = pi;            // a plain value: no engine object behind it
        edited.entropy = intToBN254(42);
        exportProofInputBN254(hashcfg, dir + "/pi_edited.json", edited);
      }
    }
    bool threw = false;
    try { hashCell(e, hashcfg, glob, Cell(63)); } catch (const AssertionDefect&) { threw = true; }
    std::printf("hashCell wrong size asserts: %s\n", threw ? "yes" : "NO");
    std::printf("%s\n", all && threw ? "ALL OK" : "SOME FAILED");
    return all && threw ? 0 : 1;
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "error: %s\n", ex.what());
    return 2;
  }
}

## OVERLAY of reference/nim/proof_input/src/blocks/bn254.nim (merkleTree, hashCell, hashNetworkBlock,
## networkBlockTree).  Uncompiled; mechanical by design.
import ../types
import ../types/bn254
import ../merkle/bn254
import ../codex_p2

proc merkleTree*(hashcfg: HashConfig, what: openArray[Hash]): MerkleTree[Hash] =   ## blocks/bn254.nim:17-19
  assert hashcfg.combo == BN254_Poseidon2
  merkleTreeBN254(what)

proc hashCell*(hashcfg: HashConfig, globcfg: GlobalConfig, cellData: Cell): Hash = ## blocks/bn254.nim:23-29
  assert hashcfg.field == BN254 and hashcfg.hashFun == Poseidon2
  assert cellData.len == globcfg.cellSize, "cells are expected to be exactly " & $globcfg.cellSize & " bytes"
  Sponge.digest(cellData, rate = 2)

proc cellHashes(hashcfg: HashConfig, globcfg: GlobalConfig, blockData: Block): seq[Hash] =
  ## all cells of one network block in ONE launch (the reference hashes them one by one, blocks/bn254.nim:33-45,56)
  assert blockData.len == globcfg.blockSize, "network blocks are expected to be exactly" & $globcfg.blockSize & " bytes"
  hashCells(blockData, globcfg.cellSize)

proc hashNetworkBlock*(hashcfg: HashConfig, globcfg: GlobalConfig, blockData: Block): Hash =   ## blocks/bn254.nim:49-54
  merkleDigestBN254(cellHashes(hashcfg, globcfg, blockData))

proc networkBlockTree*(hashcfg: HashConfig, globcfg: GlobalConfig, blockData: Block): MerkleTree[Hash] =   ## blocks/bn254.nim:60-67
  assert hashcfg.field == BN254
  merkleTree(hashcfg, cellHashes(hashcfg, globcfg, blockData))

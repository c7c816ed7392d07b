## OVERLAY of reference/nim/proof_input/src/merkle/bn254.nim (compressWithKey, merkleDigestBN254, merkleTreeBN254):
## every hash runs on the GPU through libcodex_p2.so.  Uncompiled; mechanical by design.
import ../types
import ../types/bn254
import ../codex_p2

proc compressWithkey*(key: int, x, y: F): F = compress(x, y, key = toF(key))       ## merkle/bn254.nim:18
proc merkleDigestBN254*(xs: openArray[F]): F = Merkle.digest(xs)                   ## merkle/bn254.nim:20
proc merkleTreeBN254*(xs: openArray[F]): MerkleTree[F] =                            ## merkle/bn254.nim:62-63
  ## all layers, bottom first; keys 1/0 on even layers, 3/2 for an odd tail; a singleton still gets one compression
  MerkleTree[F](layers: engineMerkleLayers(xs))

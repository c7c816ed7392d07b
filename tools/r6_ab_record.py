"""profiles/r06_streamed_files_ab.txt from the outputs of tools/r6_records.sh (one gpurun call, one box): usage r6_ab_record.py <dir>"""
import glob, json, os, re, sys
d = sys.argv[1]
order = [("ab_small_shm_r05.txt", "configs[3]'s scale-down as 4096 slot files of 8 MiB in /dev/shm (tmpfs: its pages ARE page cache) -- ROUND 5's library (commit c7434d0), CODEX_P2_LIB override"),
         ("ab_small_shm.txt", "the same files' shape, this round's library, defaults AT THAT TIME (512 MiB turns with room; 768 since: the last section, fills two turns deep, uploads on the hashing streams, passes of a group)"),
         ("ab_small_tmp.txt", "the same on the box's disk-backed /tmp (files just written: in the page cache)"),
         ("ab_small_shm_chunk256.txt", "CP2_INGEST_CHUNK_MB=256 (one wave of workgroups per launch with room)"),
         ("ab_small_shm_chunk384.txt", "CP2_INGEST_CHUNK_MB=384 (1.5 waves)"),
         ("ab_small_shm_chunk1024.txt", "CP2_INGEST_CHUNK_MB=1024 (four waves)"),
         ("ab_small_shm_copystream.txt", "CP2_INGEST_COPY_STREAM=1: uploads on a stream of their own, as until round 6 (it shares a hardware queue with the second hashing stream)"),
         ("ab_small_tmp_mapped.txt", "CP2_INGEST_MAPPED=1 on /tmp: every turn's 64 files mapped, registered by the fill threads and uploaded in place -- beside the formatting threads it loses"),
         ("ab_big_shm_r05.txt", "16 slot files of 8 GiB in /dev/shm (128 GiB) -- ROUND 5's library"),
         ("ab_big_shm.txt", "the same, this round's library"),
         ("ab_big_shm_copystream.txt", "the same with CP2_INGEST_COPY_STREAM=1")]
print("# Streamed proof inputs (cp2_dataset_build_streamed + cp2_dataset_export_streamed, every input.json formed) from REAL slot files beside the same")
print("# shape from the fake source, alternating in one process; every line of this file comes from ONE gpurun call on ONE box (tools/r6_records.sh,")
print("# tools/streamed_files_ab.py).  The files hold the reference's fake data, so both sources must give the same roots and texts: checked in every run.")
print("# " + " | ".join(l.strip() for l in open(os.path.join(d, "box.txt")).read().splitlines()))
for name, what in order:
    p = os.path.join(d, name)
    if not os.path.exists(p):
        continue
    print("\n== " + what)
    for line in open(p):
        if re.match(r"^(small|big)\s+(file|fake) run|^(small|big)\s+file/fake|^wrote", line):
            print(line.rstrip())
        elif line.startswith("{"):
            print(line.rstrip())

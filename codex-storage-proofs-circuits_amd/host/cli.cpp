// `cli` -- drop-in twin of reference/nim/proof_input/src/cli.nim for --field=bn254 --hash=poseidon2:
// same flags (cli.nim:118-154), same defaults (cli.nim:47-76), same outputs (input.json, main circom
// component), computed on the MI355X through libcodex_p2.so.  workflow/prove.sh:26 and workflow/setup.sh:13
// run it as  ${NIMCLI_DIR}/cli $CLI_ARGS -v --output=input.json  /  --circom=proof_main.circom.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "proof_input_api.hpp"

using namespace codex;

struct FullConfig {   // cli.nim:37-45
  HashConfig hashCfg;
  GlobalConfig globCfg;
  DataSetConfig dsetCfg;
  int64_t slotIndex = 0;
  int64_t entropy = 1234567;
  std::string outFile, circomFile;
  bool verbose = false;
};

static void printHelp() {   // cli.nim:80-105
  std::puts("usage:");
  std::puts("$ ./cli [options] --output=proof_input.json --circom=proof_main.circom");
  std::puts("");
  std::puts("available options:");
  std::puts(" -h, --help                         : print this help");
  std::puts(" -v, --verbose                      : verbose output (print the actual parameters)");
  std::puts(" -d, --depth      = <maxdepth>      : maximum depth of the slot tree (eg. 32)");
  std::puts(" -N, --maxslots   = <maxslots>      : maximum number of slots (eg. 256)");
  std::puts(" -c, --cellsize   = <cellSize>      : cell size in bytes (eg. 2048)");
  std::puts(" -b, --blocksize  = <blockSize>     : block size in bytes (eg. 65536)");
  std::puts(" -s, --nslots     = <nslots>        : number of slots in the dataset (eg. 13)");
  std::puts(" -n, --nsamples   = <nsamples>      : number of samples we prove (eg. 100)");
  std::puts(" -e, --entropy    = <entropy>       : external randomness (eg. 1234567)");
  std::puts(" -S, --seed       = <seed>          : seed to generate the fake data (eg. 12345)");
  std::puts(" -f, --file       = <datafile>      : slot data file, base name (eg. \"slotdata\" would mean \"slotdata5.dat\" for slot index = 5)");
  std::puts(" -i, --index      = <slotIndex>     : index of the slot (within the dataset) we prove");
  std::puts(" -k, --log2ncells = <log2(ncells)>  : log2 of the number of cells inside this slot (eg. 10)");
  std::puts(" -K, --ncells     = <ncells>        : number of cells inside this slot (eg. 1024; must be a power of two)");
  std::puts(" -o, --output     = <input.json>    : the JSON file into which we write the proof input");
  std::puts(" -C, --circom     = <main.circom>   : the circom main component to create with these parameters");
  std::puts(" -F, --field      = <field>         : the underlying field: \"bn254\" or \"goldilocks\"");
  std::puts(" -H, --hash       = <hash>          : the hash function to use: \"poseidon2\" or \"monolith\"");
  std::puts("");
  std::exit(0);
}

static std::string lower(std::string s) { for (auto& c : s) c = (char)std::tolower((unsigned char)c); return s; }
static int64_t parseInt(const std::string& v) {
  size_t pos = 0;
  long long r = 0;
  try { r = std::stoll(v, &pos, 10); } catch (...) { pos = 0; }
  if (pos == 0 || pos != v.size()) { std::cerr << "invalid integer: " << v << "\n"; std::exit(1); }
  return r;
}

// std/parseopt conventions: --key=value, --key:value, -k=value, -k:value, -kvalue; bare words are arguments
static FullConfig parseCliOptions(int argc, char** argv) {
  FullConfig full;
  // cli.nim:47-76 defaults (field default is Goldilocks in the reference)
  full.hashCfg.field = FieldSelect::Goldilocks;
  full.hashCfg.hashFun = HashSelect::Poseidon2;
  full.globCfg = GlobalConfig{32, 8, 2048, 65536};
  full.dsetCfg.nCells = 256; full.dsetCfg.nSamples = 5; full.dsetCfg.nSlots = 11;
  full.dsetCfg.dataSrc.kind = DataSourceKind::FakeData; full.dsetCfg.dataSrc.seed = 12345;
  for (int a = 1; a < argc; ++a) {
    std::string arg = argv[a], key, value;
    if (arg.rfind("--", 0) == 0) {
      size_t p = arg.find_first_of("=:", 2);
      key = arg.substr(2, p == std::string::npos ? std::string::npos : p - 2);
      if (p != std::string::npos) value = arg.substr(p + 1);
    } else if (arg.size() >= 2 && arg[0] == '-') {
      key = arg.substr(1, 1);
      value = arg.substr(2);
      if (!value.empty() && (value[0] == '=' || value[0] == ':')) value = value.substr(1);
    } else {
      continue;   // positional argument: ignored (cli.nim:122-124)
    }
    auto& g = full.globCfg;
    auto& d = full.dsetCfg;
    if (key == "h" || key == "help") printHelp();
    else if (key == "v" || key == "verbose") full.verbose = true;
    else if (key == "d" || key == "depth") g.maxDepth = parseInt(value);
    else if (key == "N" || key == "maxslots") g.maxLog2NSlots = ceilingLog2(parseInt(value));
    else if (key == "c" || key == "cellsize") g.cellSize = checkPowerOfTwo(parseInt(value), "cellSize");
    else if (key == "b" || key == "blocksize") g.blockSize = checkPowerOfTwo(parseInt(value), "blockSize");
    else if (key == "s" || key == "nslots") d.nSlots = parseInt(value);
    else if (key == "n" || key == "nsamples") d.nSamples = parseInt(value);
    else if (key == "e" || key == "entropy") full.entropy = parseInt(value);
    else if (key == "S" || key == "seed") { d.dataSrc = DataSource{}; d.dataSrc.kind = DataSourceKind::FakeData; d.dataSrc.seed = (uint64_t)parseInt(value); }
    else if (key == "f" || key == "file") { d.dataSrc = DataSource{}; d.dataSrc.kind = DataSourceKind::SlotFile; d.dataSrc.filename = value; }
    else if (key == "i" || key == "index") full.slotIndex = parseInt(value);
    else if (key == "k" || key == "log2ncells") d.nCells = pow2((int)parseInt(value));
    else if (key == "K" || key == "ncells") d.nCells = checkPowerOfTwo(parseInt(value), "nCells");
    else if (key == "o" || key == "output") full.outFile = value;
    else if (key == "C" || key == "circom") full.circomFile = value;
    else if (key == "F" || key == "field") {
      std::string f = lower(value);
      if (f == "bn254") full.hashCfg.field = FieldSelect::BN254;
      else if (f == "goldilocks") full.hashCfg.field = FieldSelect::Goldilocks;
      else throw AssertionDefect("parsefield: unrecognized field `" + value + "`");
    } else if (key == "H" || key == "hash") {
      std::string h = lower(value);
      if (h == "poseidon2") full.hashCfg.hashFun = HashSelect::Poseidon2;
      else if (h == "monolith") full.hashCfg.hashFun = HashSelect::Monolith;
      else throw AssertionDefect("parsefield: unrecognized hash function `" + value + "`");
    } else {
      std::cout << "Unknown option: " << key << "\n" << "use --help to get a list of options\n";   // cli.nim:148-151
      std::exit(0);
    }
  }
  // toFieldHashCombo, types.nim:135-148
  if (full.hashCfg.field == FieldSelect::BN254) {
    if (full.hashCfg.hashFun != HashSelect::Poseidon2) throw AssertionDefect("invalid hash function `Monolith` choice for field `BN254`");
    full.hashCfg.combo = FieldHashCombo::BN254_Poseidon2;
  } else {
    full.hashCfg.combo = full.hashCfg.hashFun == HashSelect::Poseidon2 ? FieldHashCombo::Goldilocks_Poseidon2 : FieldHashCombo::Goldilocks_Monolith;
  }
  return full;
}

static void printConfig(const FullConfig& f) {   // cli.nim:166-182
  std::cout << "field      = " << (f.hashCfg.field == FieldSelect::BN254 ? "BN254" : "Goldilocks") << "\n";
  std::cout << "hash func. = " << (f.hashCfg.hashFun == HashSelect::Poseidon2 ? "Poseidon2" : "Monolith") << "\n";
  std::cout << "maxDepth   = " << f.globCfg.maxDepth << "\n";
  std::cout << "maxSlots   = " << pow2((int)f.globCfg.maxLog2NSlots) << "\n";
  std::cout << "cellSize   = " << f.globCfg.cellSize << "\n";
  std::cout << "blockSize  = " << f.globCfg.blockSize << "\n";
  std::cout << "nSamples   = " << f.dsetCfg.nSamples << "\n";
  std::cout << "entropy    = " << f.entropy << "\n";
  std::cout << "slotIndex  = " << f.slotIndex << "\n";
  std::cout << "nCells     = " << f.dsetCfg.nCells << "\n";
  if (f.dsetCfg.dataSrc.kind == DataSourceKind::FakeData) std::cout << "dataSource = (kind: FakeData, seed: " << f.dsetCfg.dataSrc.seed << ")\n";
  else std::cout << "dataSource = (kind: SlotFile, filename: \"" << f.dsetCfg.dataSrc.filename << "\")\n";
}

static void writeCircomMainComponent(const FullConfig& f, const std::string& fname) {   // cli.nim:186-204
  exactLog2(f.globCfg.blockSize / f.globCfg.cellSize);
  cp2_config c = toEngineConfig(f.globCfg, f.dsetCfg);
  int st = cp2_write_circom_main(&c, fname.c_str());
  if (st != CP2_OK) throw std::runtime_error(std::string("writeCircomMainComponent: ") + cp2_strerror(st));
}

int main(int argc, char** argv) {
  try {
    FullConfig fullCfg = parseCliOptions(argc, argv);
    if (fullCfg.verbose) printConfig(fullCfg);
    if (fullCfg.circomFile.empty() && fullCfg.outFile.empty()) {
      std::cout << "nothing to do!\nuse --help for getting a list of options\n";
      return 0;
    }
    if (!fullCfg.circomFile.empty()) {
      std::cout << "writing circom main component into `" << fullCfg.circomFile << "`\n";
      writeCircomMainComponent(fullCfg, fullCfg.circomFile);
    }
    if (!fullCfg.outFile.empty()) {
      std::cout << "writing proof input into `" << fullCfg.outFile << "`...\n";
      if (fullCfg.hashCfg.field != FieldSelect::BN254) {
        std::cerr << "this build implements --field=bn254 --hash=poseidon2 only (the combination workflow/cli_args.sh passes); "
                     "the Goldilocks variants are out of scope\n";
        return 2;
      }
      Engine engine;   // one GPU; CODEX_P2_GPUS=all | <count> | <index list> opts in to several (no new flag: the reference's flag set stays exact)
      Entropy entropy = intToBN254(fullCfg.entropy);
      SlotProofInput prfInput = generateProofInputBN254(engine, fullCfg.hashCfg, fullCfg.globCfg, fullCfg.dsetCfg, fullCfg.slotIndex, entropy);
      exportProofInputBN254(fullCfg.hashCfg, fullCfg.outFile, prfInput);
    }
    std::cout << "done\n";
    return 0;
  } catch (const AssertionDefect& e) {
    std::cerr << "Error: unhandled exception: " << e.what() << " [AssertionDefect]\n";
    return 1;
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
}

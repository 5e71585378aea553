// cli_main.cpp -- `Ahsoka phase -g <gfa> -a <gaf> -o <prefix> [-t N]` and
// `Ahsoka only-bubbles -g <gfa> -o <prefix>`: the reference's command line
// (src/polyassembly.cpp:39-57) over this repository's ingest and HIP path.
// -t is accepted and ignored: chains are phased by the GPU, not by host threads
// (the reference's -t>1 mode only handles ten chains, quirk Q16).  -d 0,1,.. (new)
// shards the chains over several GPUs of the node (ahsoka_phase_batch_multi).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/ahsoka_phase.h"

static void usage() {
  std::cerr << "Ahsoka [options] -g <graph.gfa> -a <alignments.gaf>\n"
               "  Ahsoka phase -g <gfa> -a <gaf> -o <output prefix> [-t threads] [-d dev0,dev1,...]\n"
               "  Ahsoka only-bubbles -g <gfa> -o <output prefix>\n";
}

// <o>-bubbleinfo.txt, src/polyassembly.cpp:95-110, from the ingest dump
static int write_bubbleinfo(const std::string& gfa, const std::string& prefix) {
  const std::string tmp = prefix + "-ingest.tmp";
  int rc = ahsoka_ingest_dump(gfa.c_str(), "", tmp.c_str());
  if (rc) return rc;
  std::ifstream in(tmp);
  std::ofstream out(prefix + "-bubbleinfo.txt");
  std::string line;
  while (std::getline(in, line)) {
    std::stringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "CHAIN") {
      int id, nb;
      ss >> id >> nb;
      out << "chain id: " << id << "size: " << nb << "\n";
    } else if (tag == "BUBBLE") {
      int chain, id, src, sink, ninner;
      ss >> chain >> id >> src >> sink >> ninner;
      out << "bubble id: " << id << "\n" << "node id: " << src << ",";
      for (int i = 0; i < ninner; i++) { int x; ss >> x; out << x << ","; }
      out << sink << ",\n";
    }
  }
  in.close();
  std::remove(tmp.c_str());
  return 0;
}

int main(int argc, char** argv) {
  std::cerr << "Ahsoka (MI355X phasing path): haplotype assembly from an assembly graph and long-read graph alignments\n";
  if (argc < 2) { usage(); return 1; }
  const std::string cmd = argv[1];
  if (cmd != "phase" && cmd != "only-bubbles") { usage(); return 1; }
  std::string gfa, gaf, prefix;
  std::vector<int32_t> devices;
  for (int i = 2; i + 1 < argc; i += 2) {
    const std::string f = argv[i];
    if (f == "-g") gfa = argv[i + 1];
    else if (f == "-a") gaf = argv[i + 1];
    else if (f == "-o") prefix = argv[i + 1];
    else if (f == "-t" || f == "-s") { /* accepted, unused */ }
    else if (f == "-d") {
      std::stringstream ss(argv[i + 1]);
      std::string tok;
      while (std::getline(ss, tok, ',')) {
        char* endp = nullptr;
        const long v = std::strtol(tok.c_str(), &endp, 10);
        if (tok.empty() || *endp || v < 0 || v > 1023) { usage(); return 1; }
        devices.push_back((int32_t)v);
      }
      if (devices.empty()) { usage(); return 1; }
    }
    else { usage(); return 1; }
  }
  if (gfa.empty() || prefix.empty() || (cmd == "phase" && gaf.empty())) { usage(); return 1; }
  if (int rc = write_bubbleinfo(gfa, prefix)) { std::cerr << "cannot read graph (" << rc << ")\n"; return 1; }
  std::cout << "Step 2: Bubbles read" << std::endl;
  if (cmd == "only-bubbles") return 0;

  ahsoka_chain_batch* batch = nullptr;
  int rc = ahsoka_ingest(gfa.c_str(), gaf.c_str(), &batch);
  if (rc) { std::cerr << "ingest failed (" << rc << ")\n"; return 1; }
  std::cout << "Step 3: Alignments read" << std::endl;
  std::cout << "Number of bubble chains: " << batch->n_chains << std::endl;
  ahsoka_phase_params prm;
  ahsoka_default_params(&prm);
  ahsoka_phase_result res;
  if (devices.empty()) rc = ahsoka_phase_batch(batch, &prm, &res);
  else rc = ahsoka_phase_batch_multi(batch, &prm, devices.data(), (int32_t)devices.size(), &res);
  if (rc) {
    std::cerr << "phasing failed (" << rc << "): this build has no CPU path, a HIP device is required\n";
    ahsoka_batch_host_free(batch);
    return 1;
  }
  std::cout << "Step 5: Phasing processed (" << res.elapsed_ms << " ms)" << std::endl;
  // <o>-result.txt is opened in append mode by the reference (:72)
  std::ofstream full(prefix + "-result.txt", std::ios_base::app);
  for (int c = 0; c < batch->n_chains; c++) {
    const int64_t need = ahsoka_render_chain(batch, &res, c, nullptr, 0);
    if (need < 0) continue;
    std::vector<char> buf((size_t)need + 1);
    ahsoka_render_chain(batch, &res, c, buf.data(), need + 1);
    full << buf.data();
    if (res.status[c] == AHSOKA_ST_PHASED) {
      // <o>-chain<N>-result.txt holds the node lines only (:411-412,449,456)
      std::ofstream per(prefix + "-chain" + std::to_string(batch->chain_id[c]) + "-result.txt");
      std::stringstream ss(buf.data());
      std::string line;
      int ln = 0;
      while (std::getline(ss, line)) {
        if (ln >= 2 && line.rfind("haplotype ", 0) != 0) per << line << "\n";
        ln++;
      }
    }
  }
  ahsoka_result_free(&res);
  ahsoka_batch_host_free(batch);
  return 0;
}

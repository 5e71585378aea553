#!/usr/bin/env python3
"""Experiment support (CPU, test infrastructure): generates oracle/_exp_tmp.cpp = the oracle with a
simulation of three row-maximum policies riding on its decision sequence (they cannot change
it: all three are exact) and counts the row rescans each would do:

  EAGER     the kernel's policy: a row whose best edge is resolved is re-derived at once
  SB        second-best bounds: the row keeps an upper bound on its other edges and is re-derived
            when that bound ties the maximum
  SB-batch  the same, re-deriving every stale row whose bound reaches the largest exact maximum

    python tools/experiments/sim_rescan_policies.py   # writes oracle/_exp_tmp.cpp (git-ignored name: delete after use)
    g++ -std=c++17 -O2 -fPIC -shared -pthread -Iinclude -o /tmp/liboracle_sim.so oracle/_exp_tmp.cpp
    CE_SIM=1 python - <<< "... load /tmp/liboracle_sim.so with ctypes and call oracle_phase_batch on one chain ..."

Result of round 2 (one chain of config 3 / config 4): EAGER 29.3 k + 0.1 k rescans, SB-batch 13.3 k in
1 573 events / EAGER 62.2 k + 5.9 k, SB-batch 31.4 k in 9 744 events.  On the GPU the policy was
bit-exact and cut the rescans as simulated, but its retry events (36 k cycles each), the returning
LDS atomics of the folds and the LDS it needs (8 B per read) made it equal in time: not kept.
"""
s=open('/root/repo/oracle/oracle.cpp').read()
old="  struct Best { int64_t val; int col; };\n"
new="""  struct Best { int64_t val; int col; };
  struct G { int64_t val[2]; int col[2]; bool stale[2]; int64_t sec[2]; };
  std::vector<G> ge, gs, gb;        // EAGER (current kernel), SB (second-best bounds, rows tying the maximum), SB-batch
  std::vector<char> need_eager;
  long long cnt_eager = 0, cnt_p1_e = 0, cnt_p1_s = 0, cnt_fast = 0, cnt_iters_general = 0, cnt_p1_b = 0, ev_b = 0, rounds_b = 0, ev_s = 0;
  void exact_row(int u, G& g) {
    int64_t b[2] = {-1, -1}, s2[2] = {-1, -1}; int c[2] = {-1, -1};
    for (int x = 0; x < n; x++) {
      if (x == u || !is_live(W[ix(u, x)])) continue;
      const int64_t v[2] = {ICF[ix(u, x)], ICP[ix(u, x)]};
      for (int k = 0; k < 2; k++) {
        if (v[k] > b[k]) { s2[k] = b[k]; b[k] = v[k]; c[k] = x; }
        else if (v[k] > s2[k]) s2[k] = v[k];
      }
    }
    for (int k = 0; k < 2; k++) { g.val[k] = b[k]; g.col[k] = c[k]; g.stale[k] = false; g.sec[k] = s2[k]; }
  }
  void fold_sb(G& g, int col, const int64_t* v) {
    for (int k = 0; k < 2; k++) {
      if (g.col[k] == col && !g.stale[k]) {
        if (v[k] < g.val[k]) { g.val[k] = std::max(v[k], g.sec[k]); g.stale[k] = true; g.col[k] = -2; g.sec[k] = g.val[k]; }
        else g.val[k] = v[k];
      } else if (v[k] > g.val[k] || (v[k] == g.val[k] && !g.stale[k] && col < g.col[k])) {
        g.sec[k] = std::max(g.sec[k], g.val[k]); g.val[k] = v[k]; g.col[k] = col; g.stale[k] = false;
      } else g.sec[k] = std::max(g.sec[k], v[k]);
    }
  }
  void sim_fold(int row, int col) {
    const int64_t v[2] = {ICF[ix(row, col)], ICP[ix(row, col)]};
    for (int k = 0; k < 2; k++) {
      G& g = ge[row];
      if (g.col[k] == col && !g.stale[k]) { if (v[k] < g.val[k]) g.stale[k] = true; else g.val[k] = v[k]; }
      else if (g.col[k] == col && g.stale[k]) { if (v[k] > g.val[k]) { g.val[k] = v[k]; } }
      else if (v[k] > g.val[k] || (v[k] == g.val[k] && !g.stale[k] && col < g.col[k])) { g.val[k] = v[k]; g.col[k] = col; g.stale[k] = false; }
    }
    fold_sb(gs[row], col, v);
    fold_sb(gb[row], col, v);
  }
  void sim_die(int row, int col, bool general) {
    for (int k = 0; k < 2; k++) {
      if (ge[row].col[k] == col) { if (general) need_eager[row] = 1; }
      for (G* g : {&gs[row], &gb[row]})
        if (g->col[k] == col) { g->val[k] = g->sec[k]; g->stale[k] = true; g->col[k] = -2; }
    }
  }
"""
assert old in s; s=s.replace(old,new,1)
s=s.replace("          dirty[y] = 1; dirty[w] = 1;\n","          dirty[y] = 1; dirty[w] = 1; chg_list.push_back({y, w});\n",1)
s=s.replace("          dirty[x] = 1; dirty[w] = 1;\n","          dirty[x] = 1; dirty[w] = 1; chg_list.push_back({x, w});\n",1)
s=s.replace("    W[ix(x, y)] = W[ix(y, x)] = N;\n    dirty[x] = 1;\n","    W[ix(x, y)] = W[ix(y, x)] = N;\n    die_list.push_back({x, y});\n    dirty[x] = 1;\n",1)
s=s.replace("  void set_pair(int x, int y, int32_t N) {","  std::vector<std::pair<int,int>> chg_list, die_list; std::vector<int> need_fast;\n  void set_pair(int x, int y, int32_t N) {",1)
old="""      if (fu < 0 || pu < 0) break;  // no live edge left
      iterations++;
"""
new="""      if (fu < 0 || pu < 0) break;  // no live edge left
      iterations++;
      {
        if (ge.empty()) { ge.resize(n); gs.resize(n); gb.resize(n); need_eager.assign(n, 0); for (int u = 0; u < n; u++) { exact_row(u, ge[u]); gs[u] = ge[u]; gb[u] = ge[u]; } }
        for (int u = 0; u < n; u++) if (need_eager[u]) { exact_row(u, ge[u]); need_eager[u] = 0; cnt_eager++; }
        const int64_t tv[2] = {vf, vp};
        for (int pol = 0; pol < 2; pol++) {
          std::vector<G>& g = pol ? gs : ge;
          bool any = false;
          for (int u = 0; u < n; u++) {
            bool hit = false;
            for (int k = 0; k < 2; k++) if (g[u].stale[k] && g[u].val[k] >= tv[k] && g[u].val[k] >= 0) hit = true;
            if (hit) { exact_row(u, g[u]); (pol ? cnt_p1_s : cnt_p1_e)++; any = true; }
          }
          if (pol && any) ev_s++;
        }
        // SB-batch: every stale row whose bound reaches the largest EXACT row maximum, in rounds
        bool first = true;
        for (;;) {
          int64_t ex[2] = {-1, -1};
          for (int u = 0; u < n; u++) for (int k = 0; k < 2; k++) if (!gb[u].stale[k]) ex[k] = std::max(ex[k], gb[u].val[k]);
          int got = 0;
          for (int u = 0; u < n; u++) {
            bool hit = false;
            for (int k = 0; k < 2; k++) if (gb[u].stale[k] && gb[u].val[k] >= ex[k] && gb[u].val[k] >= 0) hit = true;
            if (hit) { exact_row(u, gb[u]); got++; }
          }
          if (!got) break;
          cnt_p1_b += got; rounds_b++;
          if (first) { ev_b++; first = false; }
        }
      }
      chg_list.clear(); die_list.clear();
      const bool sim_fast = !(vf >= vp) && members[clique[pu]].size() == 1 && members[clique[pv]].size() == 1;
"""
assert old in s; s=s.replace(old,new,1)
old="""      if (vf >= vp) merge(fu, fv);
      else forbid(pu, pv);"""
new="""      if (vf >= vp) merge(fu, fv);
      else forbid(pu, pv);
      {
        for (auto& d : die_list) { sim_die(d.first, d.second, !sim_fast); sim_die(d.second, d.first, !sim_fast); }
        std::sort(chg_list.begin(), chg_list.end()); chg_list.erase(std::unique(chg_list.begin(), chg_list.end()), chg_list.end());
        for (auto& c : chg_list) if (is_live(W[ix(c.first, c.second)])) { sim_fold(c.first, c.second); sim_fold(c.second, c.first); }
        if (sim_fast) { cnt_fast += 2; need_fast.push_back(pu); need_fast.push_back(pv); } else cnt_iters_general++;
        for (int u : need_fast) { exact_row(u, ge[u]); gs[u] = ge[u]; gb[u] = ge[u]; }
        need_fast.clear();
      }"""
assert old in s; s=s.replace(old,new,1)
old="    // clusters ordered by smallest member, members ascending\n"
new="""    if (getenv("CE_SIM")) fprintf(stderr, "SIM n=%d iters=%lld general=%lld | fast-path row scans=%lld | EAGER: eager rescans=%lld + P1 rescans=%lld | SB: P1 rescans=%lld in %lld iterations | SB-batch: rescans=%lld in %lld iterations, %lld rounds\\n", n, (long long)iterations, cnt_iters_general, cnt_fast, cnt_eager, cnt_p1_e, cnt_p1_s, ev_s, cnt_p1_b, ev_b, rounds_b);
    // clusters ordered by smallest member, members ascending\n"""
assert old in s; s=s.replace(old,new,1)
open('/root/repo/oracle/_exp_tmp.cpp','w').write(s)

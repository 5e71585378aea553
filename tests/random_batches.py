"""Random ChainBatches with irregular structure, for differential tests: bubbles
with 1..5 alleles, allele paths of 2..5 nodes (2-node paths have an empty middle:
quirk Q6), reads that skip bubbles, leave the chain (-1 nodes), revisit nodes,
repeated read names, identities on both sides of the filter."""
import numpy as np

from ahsoka_amd import capi


def random_batch(seed, n_chains=4, max_bubbles=14, max_reads=120, max_alleles=5, dup_paths=False):
    rng = np.random.default_rng(seed)
    A = {k: [] for k in ("chain_id", "apath_node", "node_id", "rnode", "read_mapq", "read_name", "read_flags", "s1_order")}
    off = {k: [0] for k in ("bubble_off", "allele_off", "apath_off", "node_off", "read_off", "rnode_off")}
    next_id = 1
    for c in range(n_chains):
        A["chain_id"].append(c)
        B = int(rng.integers(0, max_bubbles + 1))
        # nodes: backbone s_0..s_B plus inner nodes per allele
        nodes = list(range(B + 1))  # local indices of backbone nodes
        n_local = B + 1
        paths_of = []
        for k in range(B):
            na = int(rng.integers(1, max_alleles + 1))
            paths = []
            for a in range(na):
                ninner = int(rng.integers(0, 4))  # 0 inner nodes => 2-node path
                inner = list(range(n_local, n_local + ninner))
                n_local += ninner
                p = [k] + inner + [k + 1]
                if dup_paths and inner and rng.random() < 0.3:
                    p = [k] + inner + [inner[0]] + [k + 1]  # the path visits a node twice (multiset containment)
                if rng.random() < 0.3:
                    p = p[::-1]  # sink -> source, like the reference's DFS paths (quirk Q7)
                paths.append(p)
            paths_of.append(paths)
            for p in paths:
                A["apath_node"].extend(p)
                off["apath_off"].append(len(A["apath_node"]))
            off["allele_off"].append(len(off["apath_off"]) - 1)
        off["bubble_off"].append(len(off["allele_off"]) - 1)
        A["s1_order"].extend(rng.permutation(B).tolist())  # any visiting order is legal input
        A["node_id"].extend(range(next_id, next_id + n_local))
        next_id += n_local
        off["node_off"].append(len(A["node_id"]))
        R = int(rng.integers(0, max_reads + 1)) if B > 0 else int(rng.integers(0, 5))
        names = []
        truth = [int(rng.integers(0, len(paths_of[k]))) for k in range(B)]
        alt = [int(rng.integers(0, len(paths_of[k]))) for k in range(B)]
        for r in range(R):
            if B == 0:
                path = [0]
            else:
                u = int(rng.integers(0, B))
                span = int(min(B - u, 1 + rng.geometric(0.3)))
                hap = truth if rng.random() < 0.5 else alt
                path = []
                for k in range(u, u + span):
                    a = hap[k] if rng.random() > 0.05 else int(rng.integers(0, len(paths_of[k])))
                    p = paths_of[k][a]
                    p = p if p[0] == k else p[::-1]
                    if rng.random() < 0.1:
                        p = [p[0], p[-1]]  # the read jumps over the inner nodes
                    if dup_paths and rng.random() < 0.3 and len(p) > 3:
                        p = [q for q in p if q != p[1]] + [p[1]] if rng.random() < 0.5 else p  # sometimes only one visit
                    path.extend(p if not path else p[1:])
                if rng.random() < 0.1:
                    path.insert(int(rng.integers(0, len(path) + 1)), -1)  # node of another chain
                if rng.random() < 0.05:
                    path.append(path[0])  # revisits a node
            A["rnode"].extend(path)
            off["rnode_off"].append(len(A["rnode"]))
            mq = int(rng.integers(88, 101))
            A["read_mapq"].append(mq)
            # float(id)*100 > 90: implied by the truncated value except at exactly 90
            A["read_flags"].append(int(mq > 90 or (mq == 90 and rng.random() < 0.5)))
            if names and rng.random() < 0.1:
                names.append(names[int(rng.integers(0, len(names)))])
            else:
                names.append(max(names) + 1 if names else 0)
        A["read_name"].extend(names)
        off["read_off"].append(len(A["read_mapq"]))
    arrays = {k: np.asarray(v, dtype=np.uint8 if k == "read_flags" else np.int32) for k, v in A.items()}
    arrays.update({k: np.asarray(v, dtype=np.int64) for k, v in off.items()})
    return capi.ChainBatch(arrays=arrays)

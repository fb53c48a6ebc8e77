// TEST-ONLY host build of the engine's per-lane code (csrc/bmo_lane.hpp).
//
// Purpose: (1) debug/sanitize the lane arithmetic on the CPU (GPU ASan is not available on the
// pool), (2) check lane code == oracle without spending GPU minutes.  It walks the same
// bounce-synchronous schedule as step_kernel, one "lane" at a time.  It is NOT a fallback: the
// product package never loads it (see beamletoptics.jl_amd/abi.py: the engine is libbmo_hip.so only).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../beamletoptics.jl_amd/csrc/bmo_lane.hpp"

using namespace bmo;

namespace {
struct Rec {
    RayS ray;
    Hit X;
    int node, k, hobj, hshape, flags;
    double opl;
};
struct NodeE {
    int root, parent, nseg, status, li, hit_det;
    unsigned long long key;
    double lambda, hit[9];
    int old = -1;  // node of the previous solution this beam re-walks (retrace), -1 = fresh
};
struct BeamLimit {};
// reference order of the beam nodes: bundle order x breadth-first order of each root's tree (the two children of a splitting beam are
// created consecutively, transmitted first).  Any tree depth.
template <class N>
void bfs_order(const std::vector<N>& nodes, int64_t n_roots, std::vector<int>& order) {
    const int64_t nn = (int64_t)nodes.size();
    std::vector<int> first_kid(nn, -1);
    for (int64_t c = 0; c < nn; ++c)
        if (nodes[c].parent >= 0 && first_kid[nodes[c].parent] < 0) first_kid[nodes[c].parent] = (int)c;
    order.clear();
    order.reserve(nn);
    for (int64_t r = 0; r < n_roots; ++r) {
        size_t head = order.size();
        order.push_back((int)r);
        while (head < order.size()) {
            const int k = first_kid[order[head++]];
            if (k >= 0) {
                order.push_back(k);
                order.push_back(k + 1);
            }
        }
    }
}

// node index of every root beam in a canonical result view (roots are the nodes without parent, in bundle order)
std::vector<int> old_roots(const bmo_trace_result_view* prev) {
    std::vector<int> r;
    for (int64_t i = 0; i < prev->n_nodes; ++i)
        if (prev->node_parent[i] < 0) r.push_back((int)i);
    return r;
}
struct ResultE {
    std::vector<int32_t> root, parent, first_child, first_rec, nseg, status, rec_obj, rec_shape, det_node;
    std::vector<double> aux, rec, det;
    std::vector<int64_t> det_count, det_offset;
};

template <int KIND>
void run(const bmo_scene_desc* d, const bmo_ray_batch* in, const bmo_trace_opts* opts, ResultE& R, bmo_trace_result_view* v,
         const bmo_trace_result_view* prev = nullptr) {
    std::vector<int> oroot;
    if (prev) oroot = old_roots(prev);
    SceneView S;
    S.objects = d->objects;
    S.shapes = d->shapes;
    S.children = d->children;
    S.tris = d->tris;
    S.n_table = d->n_table;
    S.coefs = d->coefs;
    S.n_objects = d->n_objects;
    S.n_lambda = d->n_lambda;
    std::vector<Cand> cands((size_t)std::max(1, fill_candidates(d->objects, d->n_objects, d->shapes, nullptr)) + 3);  // (+ 3: read four per trip)
    S.n_cands = fill_candidates(d->objects, d->n_objects, d->shapes, cands.data());
    S.cands = cands.data();
    S.eps_srf = d->eps_srf;
    S.eps_ray = d->eps_ray;
    S.eps_ins = d->eps_ins;
    S.mt_keps = d->mt_keps;
    S.mt_leps = d->mt_leps;
    S.grad_h = d->grad_h;
    S.march_iters = d->march_iters;
    const int64_t n = in->n;
    const double* P = in->planes;
    std::vector<NodeE> nodes(n);
    std::vector<Rec> cur(n), all;
    for (int64_t j = 0; j < n; ++j) {
        Rec& r = cur[j];
        r.ray.pos = {P[0 * n + j], P[1 * n + j], P[2 * n + j]};
        r.ray.dir = {P[3 * n + j], P[4 * n + j], P[5 * n + j]};
        r.ray.n = P[7 * n + j];
        if (KIND == BMO_BEAM_POLARIZED)
            for (int c = 0; c < 3; ++c) r.ray.E0[c] = {P[(8 + 2 * c) * n + j], P[(9 + 2 * c) * n + j]};
        r.node = (int)j;
        r.k = 0;
        r.hobj = r.hshape = -1;
        r.flags = (1 < opts->r_max) ? 0 : 1;
        r.opl = 0;
        nodes[j] = NodeE{(int)j, -1, 1, 0, in->lambda_idx[j], -1, 0ull, P[6 * n + j], {0}};
        if (prev) {
            nodes[j].old = oroot[j];
            r.flags = 0;  // the retrace walk ignores r_max (System.jl:197)
        }
    }
    unsigned long long calls = 0;
    int steps = 0;
    while (!cur.empty()) {
#if defined(BMO_EMU_STATS)
        const long a0 = g_emu_sdf_any, l0 = g_emu_sdf_leaf, n0 = g_emu_normal, f0 = g_emu_normal_fd;
        const size_t m0 = cur.size();
#endif
        std::vector<Rec> surv, kids;
#if defined(BMO_EMU_STATS)
        std::vector<long> per_rec;  // leaf sdf evaluations of every record of this level (wave imbalance = max over 64 records vs mean)
        per_rec.reserve(cur.size());
#endif
        for (Rec& r : cur) {
#if defined(BMO_EMU_STATS)
            struct Tally {
                std::vector<long>& v;
                long at;
                const Rec& r;
                ~Tally() {
                    v.push_back(g_emu_sdf_leaf - at);
                    if (g_emu_sdf_leaf - at > 500)
                        fprintf(stderr, "  straggler: %ld leaf evals, k %d pos %.6f %.6f %.6f dir %.6f %.6f %.6f hint %d/%d -> hit obj %d shape %d t %.6g\n", g_emu_sdf_leaf - at, r.k,
                                r.ray.pos.x, r.ray.pos.y, r.ray.pos.z, r.ray.dir.x, r.ray.dir.y, r.ray.dir.z, r.hobj, r.hshape, r.X.obj, r.X.shape, r.X.t);
                }
            } tally{per_rec, g_emu_sdf_leaf, r};
#endif
            StepOut o;
            o.outcome = OUT_MISS;
            o.status = 0;
            o.hint_obj = o.hint_shape = -1;
            o.det_slot = -1;
            double det9[9] = {0};
            o.det = det9;
            uint32_t c = 0;
            int status = 0;
            bool survive = false;
            Hit X;
            X.shape = -1;
            X.obj = -1;
            X.t = kinf();
            X.n = {0, 0, 0};
            NodeE nd = nodes[r.node];
            double opl_next = 0;
            // retrace context (System.jl:188-255)
            const int old = nd.old;
            bool probe = false, fresh_allowed = true, missed = false;
            int probe_obj = -1, old_n = 0;
            int hobj = r.hobj, hshape = r.hshape;
            if (old >= 0) {
                old_n = prev->node_nseg[old];
                probe_obj = prev->rec_obj[prev->node_first_rec[old] + r.k];
                fresh_allowed = r.k + 1 < opts->r_max;
                probe = probe_obj >= 0;
                if (!probe) {  // stored ray without intersection: cleanup, trace_system! goes on without a hint
                    missed = true;
                    hobj = hshape = -1;
                }
            }
            if ((r.flags & 1) || (old >= 0 && !probe && !fresh_allowed)) {
                status = BMO_NODE_RMAX;
            } else {
                double ccv[BMO_CC_MAX];
                ChildCache cc{ccv, 1, 0};
                double lmv[BMO_LANE_MEM];
                const LaneMem lm{lmv, 1};
                X = tracing_step<2, true>(S, r.ray.pos, r.ray.dir, hobj, hshape, c, cc, lm, probe, probe_obj, fresh_allowed, &missed);
                if (X.shape < 0) status = (old >= 0 && missed && !fresh_allowed) ? BMO_NODE_RMAX : BMO_NODE_MISS;
                else {
                    interact<KIND>(S, r.ray, X, nd.li, nd.lambda, r.opl, o, NextInOut{o});
                    status = o.status;
                    opl_next = r.opl + X.t * r.ray.n;
                    if (o.outcome == OUT_CONTINUE) survive = true;
                    else if (o.outcome == OUT_SPLIT) status |= BMO_NODE_SPLIT | BMO_NODE_STOPPED;
                    else status |= BMO_NODE_STOPPED;
                }
            }
            calls += c;
            r.X = X;
            const bool still = old >= 0 && probe && !missed;  // the stored path held at this ray
            const bool old_kids = still && prev->node_first_child[old] >= 0;
            // the re-walk ends in a `nothing` interaction before the stored splitter: the reference keeps the (stale) children and retraces
            // each from its stored first ray (System.jl:232-240, 446-458)
            const bool stale_kids = !survive && old_kids && o.outcome != OUT_SPLIT && X.shape >= 0;
            if (!survive) {
                if (stale_kids) status |= BMO_NODE_RETRACE_STALE;
                nodes[r.node].nseg = r.k + 1;
                nodes[r.node].status = status;
                if (o.det_slot >= 0) {
                    nodes[r.node].hit_det = o.det_slot;
                    std::memcpy(nodes[r.node].hit, det9, sizeof det9);
                }
            }
            if (survive) {
                Rec q;
                q.ray = o.next;
                q.node = r.node;
                q.k = r.k + 1;
                q.hobj = o.hint_obj;
                q.hshape = o.hint_shape;
                q.flags = (r.k + 2 < opts->r_max) ? 0 : 1;
                if (still && r.k + 1 < old_n) {
                    q.flags = 0;  // replace!: the next stored ray is re-walked whatever r_max says
                } else {
                    nodes[r.node].old = -1;
                    if (still) q.hobj = q.hshape = -1;  // push!, then trace_system! starts over without a hint
                }
                q.opl = opl_next;
                surv.push_back(q);
            } else {
                nodes[r.node].old = -1;
            }
            if ((o.outcome == OUT_SPLIT || stale_kids) && !(r.flags & 1) && X.shape >= 0) {
                const int root = nodes[r.node].root;
                for (int w = 0; w < 2; ++w) {
                    NodeE c2{root, r.node, 1, 0, nd.li, -1, (unsigned long long)w, nd.lambda, {0}};
                    if (old_kids) c2.old = prev->node_first_child[old] + w;  // children!: _modify_beam_head! of the stored child
                    nodes.push_back(c2);
                    Rec q;
                    q.ray = w == 0 ? o.next : o.refl;
                    if (stale_kids) {  // the stored first ray of the kept child
                        const int64_t NR = prev->n_records, at = prev->node_first_rec[c2.old];
                        const double* P = prev->rec;
                        q.ray.pos = {P[0 * NR + at], P[1 * NR + at], P[2 * NR + at]};
                        q.ray.dir = {P[3 * NR + at], P[4 * NR + at], P[5 * NR + at]};
                        q.ray.n = P[6 * NR + at];
                        if (KIND == BMO_BEAM_POLARIZED)
                            for (int cc3 = 0; cc3 < 3; ++cc3) q.ray.E0[cc3] = {P[(11 + 2 * cc3) * NR + at], P[(12 + 2 * cc3) * NR + at]};
                    }
                    q.node = (int)nodes.size() - 1;
                    q.k = 0;
                    q.hobj = q.hshape = -1;
                    q.flags = (c2.old >= 0 || 1 < opts->r_max) ? 0 : 1;
                    q.opl = opl_next;
                    kids.push_back(q);
                }
            }
        }
        all.insert(all.end(), cur.begin(), cur.end());
#if defined(BMO_EMU_STATS)
        fprintf(stderr, "step %2d records %6zu  union sdf evals/record %7.2f  leaf sdf evals/record %7.2f  dual normals/record %5.2f  numeric fallbacks/record %5.2f\n",
                steps, m0, double(g_emu_sdf_any - a0) / m0, double(g_emu_sdf_leaf - l0) / m0, double(g_emu_normal - n0) / m0, double(g_emu_normal_fd - f0) / m0);
        {
            long wave_max_sum = 0, mx = 0;
            for (size_t w = 0; w < per_rec.size(); w += 64) {
                long m = 0;
                for (size_t q = w; q < std::min(per_rec.size(), w + 64); ++q) m = std::max(m, per_rec[q]);
                wave_max_sum += m * 64;
                mx = std::max(mx, m);
            }
            std::vector<long> srt = per_rec;
            std::sort(srt.begin(), srt.end());
            fprintf(stderr, "         leaf evals per record: p50 %ld  p90 %ld  p99 %ld  max %ld   lane utilisation if a wave waits for its slowest lane: %.1f %%\n",
                    srt[srt.size() / 2], srt[srt.size() * 9 / 10], srt[srt.size() * 99 / 100], mx, 100.0 * double(g_emu_sdf_leaf - l0) / double(std::max(1L, wave_max_sum)));
        }
#endif
        cur = surv;
        cur.insert(cur.end(), kids.begin(), kids.end());
        steps += 1;
        if (opts->max_beams > 0 && (int64_t)nodes.size() > (int64_t)opts->max_beams) throw BeamLimit{};  // bmo_trace_opts.max_beams, like the engine
    }
    // canonical order
    const int64_t nn = (int64_t)nodes.size(), nr = (int64_t)all.size();
    std::vector<int> order(nn), rank(nn);
    bfs_order(nodes, n, order);
    for (int64_t i = 0; i < nn; ++i) rank[order[i]] = (int)i;
    const int PL = KIND == BMO_BEAM_RAY ? 11 : 17;
    R.root.resize(nn);
    R.parent.resize(nn);
    R.first_child.assign(nn, -1);
    R.first_rec.resize(nn);
    R.nseg.resize(nn);
    R.status.resize(nn);
    R.aux.assign(nn * 4, 0.0);
    int64_t acc = 0;
    for (int64_t i = 0; i < nn; ++i) {
        const NodeE& nd = nodes[order[i]];
        R.root[i] = nd.root;
        R.parent[i] = nd.parent < 0 ? -1 : rank[nd.parent];
        R.nseg[i] = nd.nseg;
        R.status[i] = nd.status;
        R.first_rec[i] = (int32_t)acc;
        R.aux[4 * i] = nd.lambda;
        acc += nd.nseg;
    }
    for (int64_t i = 0; i < nn; ++i)
        if (R.parent[i] >= 0 && (R.first_child[R.parent[i]] < 0 || i < R.first_child[R.parent[i]])) R.first_child[R.parent[i]] = (int32_t)i;
    R.rec.assign((size_t)PL * nr, 0.0);
    R.rec_obj.assign(nr, -1);
    R.rec_shape.assign(nr, -1);
    for (const Rec& r : all) {
        int64_t dst = (int64_t)R.first_rec[rank[r.node]] + r.k;
        double vals[17] = {r.ray.pos.x, r.ray.pos.y, r.ray.pos.z, r.ray.dir.x, r.ray.dir.y, r.ray.dir.z, r.ray.n, r.X.t, r.X.n.x, r.X.n.y, r.X.n.z,
                           r.ray.E0[0].re, r.ray.E0[0].im, r.ray.E0[1].re, r.ray.E0[1].im, r.ray.E0[2].re, r.ray.E0[2].im};
        for (int p = 0; p < PL; ++p) R.rec[(size_t)p * nr + dst] = vals[p];
        R.rec_obj[dst] = r.X.obj;
        R.rec_shape[dst] = r.X.shape;
    }
    const int ndet = d->n_detectors;
    R.det_count.assign(ndet, 0);
    R.det_offset.assign(ndet, 0);
    for (int q = 0; q < ndet; ++q) {
        R.det_offset[q] = (int64_t)R.det_node.size();
        for (int64_t i = 0; i < nn; ++i) {
            const NodeE& nd = nodes[order[i]];
            if (nd.hit_det != q) continue;
            R.det_node.push_back((int32_t)i);
            R.det.insert(R.det.end(), nd.hit, nd.hit + 9);
        }
        R.det_count[q] = (int64_t)R.det_node.size() - R.det_offset[q];
    }
    std::memset(v, 0, sizeof *v);
    v->n_roots = n;
    v->n_nodes = nn;
    v->n_records = nr;
    v->n_intersect_calls = (int64_t)calls;
    v->n_steps = steps;
    v->beam_kind = KIND;
    v->rec_planes = PL;
    v->n_detectors = ndet;
    v->node_root = R.root.data();
    v->node_parent = R.parent.data();
    v->node_first_child = R.first_child.data();
    v->node_first_rec = R.first_rec.data();
    v->node_nseg = R.nseg.data();
    v->node_status = R.status.data();
    v->node_aux = R.aux.data();
    v->rec_obj = R.rec_obj.data();
    v->rec_shape = R.rec_shape.data();
    v->rec = R.rec.data();
    v->det_count = R.det_count.data();
    v->det_offset = R.det_offset.data();
    v->det_node = R.det_node.data();
    v->det_data = R.det.data();
}

// ---- GaussianBeamlet schedule (mirrors step_kernel_gauss) ----------------------------------------------------
struct GRec {
    GaussIn g;
    GaussOut o;
    int node, k, flags;
};
struct GNode {
    int root, parent, nseg, status, li, hit_det;
    unsigned long long key;
    double lambda, l0, w0;
    cx E0;
    double hit[27];
    int old = -1;
};
// the stored rays behind the one a re-walking beamlet is at (bmo_lane.hpp gauss_step_rec `tail`), read from the previous solution's view
struct EmuTail {
    const bmo_trace_result_view* prev;
    int old, k, old_n;
    int more() const { return (prev && old >= 0) ? old_n - (k + 1) : 0; }
    double t(int q) const { return prev->rec[(int64_t)7 * prev->n_records + prev->node_first_rec[old] + k + 1 + q]; }
    RayS ray(int q, int r) const {
        const int64_t NR = prev->n_records, at = prev->node_first_rec[old] + k + 1 + q;
        const double* P = prev->rec + (int64_t)11 * r * NR;
        RayS x;
        x.pos = {P[0 * NR + at], P[1 * NR + at], P[2 * NR + at]};
        x.dir = {P[3 * NR + at], P[4 * NR + at], P[5 * NR + at]};
        x.n = P[6 * NR + at];
        return x;
    }
};
void run_gauss(const bmo_scene_desc* d, const bmo_ray_batch* in, const bmo_trace_opts* opts, ResultE& R, bmo_trace_result_view* v,
               const bmo_trace_result_view* prev = nullptr) {
    std::vector<int> oroot;
    if (prev) oroot = old_roots(prev);
    SceneView S;
    S.objects = d->objects;
    S.shapes = d->shapes;
    S.children = d->children;
    S.tris = d->tris;
    S.n_table = d->n_table;
    S.coefs = d->coefs;
    S.n_objects = d->n_objects;
    S.n_lambda = d->n_lambda;
    std::vector<Cand> cands((size_t)std::max(1, fill_candidates(d->objects, d->n_objects, d->shapes, nullptr)) + 3);  // (+ 3: read four per trip)
    S.n_cands = fill_candidates(d->objects, d->n_objects, d->shapes, cands.data());
    S.cands = cands.data();
    S.eps_srf = d->eps_srf;
    S.eps_ray = d->eps_ray;
    S.eps_ins = d->eps_ins;
    S.mt_keps = d->mt_keps;
    S.mt_leps = d->mt_leps;
    S.grad_h = d->grad_h;
    S.march_iters = d->march_iters;
    const int64_t n = in->n;
    const double* P = in->planes;
    std::vector<GNode> nodes(n);
    std::vector<GRec> cur(n), all;
    auto mk = [&](int base, int64_t j, double nn) {
        RayS r;
        r.pos = {P[(base + 0) * n + j], P[(base + 1) * n + j], P[(base + 2) * n + j]};
        r.dir = {P[(base + 3) * n + j], P[(base + 4) * n + j], P[(base + 5) * n + j]};
        r.n = nn;
        return r;
    };
    for (int64_t j = 0; j < n; ++j) {
        GRec& r = cur[j];
        double nn = P[19 * n + j];
        r.g.c = mk(0, j, nn);
        r.g.w = mk(6, j, nn);
        r.g.d = mk(12, j, nn);
        r.g.hint_obj = r.g.hint_shape = -1;
        r.g.lenA = r.g.lenB = r.g.l0 = r.g.oplC = r.g.oplW = r.g.oplD = 0.0;
        r.g.lambda = P[18 * n + j];
        r.g.w0 = P[20 * n + j];
        r.g.E0 = {P[21 * n + j], P[22 * n + j]};
        r.g.li = in->lambda_idx[j];
        r.node = (int)j;
        r.k = 0;
        r.flags = (1 < opts->r_max) ? 0 : 1;
        GNode nd{};
        nd.root = (int)j;
        nd.parent = -1;
        nd.nseg = 1;
        nd.li = r.g.li;
        nd.hit_det = -1;
        nd.key = 0;
        nd.lambda = r.g.lambda;
        nd.w0 = r.g.w0;
        nd.E0 = r.g.E0;
        if (prev) {
            nd.old = oroot[j];
            r.flags = 0;
        }
        nodes[j] = nd;
    }
    unsigned long long calls = 0;
    int steps = 0;
    while (!cur.empty()) {
        std::vector<GRec> surv, kids;
        for (GRec& r : cur) {
            uint32_t c = 0;
            int status = 0;
            bool survive = false;
            r.o.Xc = r.o.Xw = r.o.Xd = no_hit();
            r.o.outcome = OUT_MISS;
            r.o.det_slot = -1;
            r.o.det = nodes[r.node].hit;
            const int old = nodes[r.node].old;
            bool probe = false, fresh_allowed = true, missed = false;
            int probe_obj = -1, old_n = 0;
            if (old >= 0) {
                old_n = prev->node_nseg[old];
                probe_obj = prev->rec_obj[prev->node_first_rec[old] + r.k];
                fresh_allowed = r.k + 1 < opts->r_max;
                probe = probe_obj >= 0;
                if (!probe) {
                    missed = true;
                    r.g.hint_obj = r.g.hint_shape = -1;
                }
            }
            if ((r.flags & 1) || (old >= 0 && !probe && !fresh_allowed)) status = BMO_NODE_RMAX;
            else {
                double ccv[BMO_CC_MAX];
                ChildCache cc{ccv, 1, 0};
                double lmv[BMO_LANE_MEM];
                const LaneMem lm{lmv, 1};
                const EmuTail tail{prev, old, r.k, old_n};
                gauss_step<2, true, EmuTail>(S, r.g, r.o, c, cc, lm, probe, probe_obj, fresh_allowed, &missed, tail);
                status = r.o.status;
                if (r.o.outcome == OUT_CONTINUE) survive = true;
                else if (r.o.outcome == OUT_SPLIT) status |= BMO_NODE_SPLIT | BMO_NODE_STOPPED;
                else if (r.o.outcome == OUT_STOP) status |= BMO_NODE_STOPPED;
            }
            calls += c;
            const bool still = old >= 0 && probe && !missed;
            const bool old_kids = still && prev->node_first_child[old] >= 0;
            const bool keep_walking = survive && still && r.k + 1 < old_n;
            if (!keep_walking) nodes[r.node].old = -1;
            // a `nothing` interaction before the stored splitter: the children are kept and re-walk from their stored first rays (System.jl:393-400)
            const bool stale_kids = !survive && old_kids && r.o.outcome != OUT_SPLIT && !(r.flags & 1);
            if (!survive) {
                if (stale_kids) status |= BMO_NODE_RETRACE_STALE;
                if (r.o.outcome == OUT_SPLIT && still && r.k + 1 < old_n) status |= BMO_NODE_RETRACE_STALE;  // the reference sizes the children with the stale tail (gauss_step_rec `tail`)
                nodes[r.node].nseg = r.k + 1;
                nodes[r.node].status = status;
                if (r.o.det_slot >= 0 && !(r.flags & 1)) nodes[r.node].hit_det = r.o.det_slot;
            }
            auto next = [&](const RayS& c1, const RayS& w1, const RayS& d1, int node, int k, int ho, int hs, int fl, double lenA, double lenB,
                            double oplC, double oplW, double oplD, const GNode& nd) {
                GRec q;
                q.g = r.g;
                q.g.c = c1;
                q.g.w = w1;
                q.g.d = d1;
                q.g.hint_obj = ho;
                q.g.hint_shape = hs;
                q.g.lenA = lenA;
                q.g.lenB = lenB;
                q.g.oplC = oplC;
                q.g.oplW = oplW;
                q.g.oplD = oplD;
                q.g.l0 = nd.l0;
                q.g.w0 = nd.w0;
                q.g.E0 = nd.E0;
                q.node = node;
                q.k = k;
                q.flags = fl;
                return q;
            };
            if (survive) {
                const bool pushed = still && !keep_walking;  // push!, then trace_system! starts over without a hint
                surv.push_back(next(r.o.nc, r.o.nw, r.o.nd, r.node, r.k + 1, pushed ? -1 : r.o.hint_obj, pushed ? -1 : r.o.hint_shape,
                                    (keep_walking || r.k + 2 < opts->r_max) ? 0 : 1, r.o.lenA, r.o.lenB, r.o.oplC, r.o.oplW, r.o.oplD, nodes[r.node]));
            }
            if (stale_kids) {
                const int oc = prev->node_first_child[old];
                const int64_t NR = prev->n_records;
                auto head = [&](int node, int rr) {
                    const int64_t at = prev->node_first_rec[node];
                    const double* P = prev->rec + (int64_t)11 * rr * NR;
                    RayS x;
                    x.pos = {P[0 * NR + at], P[1 * NR + at], P[2 * NR + at]};
                    x.dir = {P[3 * NR + at], P[4 * NR + at], P[5 * NR + at]};
                    x.n = P[6 * NR + at];
                    return x;
                };
                r.o.nc = head(oc, 0), r.o.nw = head(oc, 1), r.o.nd = head(oc, 2);
                r.o.rc = head(oc + 1, 0), r.o.rw = head(oc + 1, 1), r.o.rd = head(oc + 1, 2);
                r.o.child_l0 = r.o.lenA + nodes[r.node].l0;
                r.o.child_w0 = prev->node_aux[4 * oc + 0];
                r.o.Et = {prev->node_aux[4 * oc + 1], prev->node_aux[4 * oc + 2]};
                r.o.Er = {prev->node_aux[4 * (oc + 1) + 1], prev->node_aux[4 * (oc + 1) + 2]};
            }
            if (!(r.flags & 1) && (r.o.outcome == OUT_SPLIT || stale_kids)) {
                const int root = nodes[r.node].root;
                for (int w = 0; w < 2; ++w) {
                    GNode c2{};
                    c2.root = root;
                    c2.parent = r.node;
                    c2.nseg = 1;
                    c2.li = r.g.li;
                    c2.hit_det = -1;
                    c2.key = (unsigned long long)w;
                    c2.lambda = r.g.lambda;
                    c2.l0 = r.o.child_l0;
                    c2.w0 = r.o.child_w0;
                    c2.E0 = w == 0 ? r.o.Et : r.o.Er;
                    if (old_kids) {
                        c2.old = prev->node_first_child[old] + w;
                        c2.w0 = prev->node_aux[4 * c2.old + 0];  // _modify_beam_head! (Gaussian.jl:154-161) leaves w0 alone
                    }
                    nodes.push_back(c2);
                    const int fl = (c2.old >= 0 || 1 < opts->r_max) ? 0 : 1;
                    if (w == 0) kids.push_back(next(r.o.nc, r.o.nw, r.o.nd, (int)nodes.size() - 1, 0, -1, -1, fl, 0.0, r.o.child_l0, r.o.oplC, 0.0, 0.0, c2));
                    else kids.push_back(next(r.o.rc, r.o.rw, r.o.rd, (int)nodes.size() - 1, 0, -1, -1, fl, 0.0, r.o.child_l0, r.o.oplC, 0.0, 0.0, c2));
                }
            }
        }
        all.insert(all.end(), cur.begin(), cur.end());
        cur = surv;
        cur.insert(cur.end(), kids.begin(), kids.end());
        steps += 1;
        if (opts->max_beams > 0 && (int64_t)nodes.size() > (int64_t)opts->max_beams) throw BeamLimit{};  // bmo_trace_opts.max_beams, like the engine
    }
    const int64_t nn = (int64_t)nodes.size(), nr = (int64_t)all.size();
    std::vector<int> order(nn), rank(nn);
    bfs_order(nodes, n, order);
    for (int64_t i = 0; i < nn; ++i) rank[order[i]] = (int)i;
    const int PL = 33;
    R.root.resize(nn);
    R.parent.resize(nn);
    R.first_child.assign(nn, -1);
    R.first_rec.resize(nn);
    R.nseg.resize(nn);
    R.status.resize(nn);
    R.aux.assign(nn * 4, 0.0);
    int64_t acc = 0;
    for (int64_t i = 0; i < nn; ++i) {
        const GNode& nd = nodes[order[i]];
        R.root[i] = nd.root;
        R.parent[i] = nd.parent < 0 ? -1 : rank[nd.parent];
        R.nseg[i] = nd.nseg;
        R.status[i] = nd.status;
        R.first_rec[i] = (int32_t)acc;
        R.aux[4 * i + 0] = nd.w0;
        R.aux[4 * i + 1] = nd.E0.re;
        R.aux[4 * i + 2] = nd.E0.im;
        R.aux[4 * i + 3] = nd.lambda;
        acc += nd.nseg;
    }
    for (int64_t i = 0; i < nn; ++i)
        if (R.parent[i] >= 0 && (R.first_child[R.parent[i]] < 0 || i < R.first_child[R.parent[i]])) R.first_child[R.parent[i]] = (int32_t)i;
    R.rec.assign((size_t)PL * nr, 0.0);
    R.rec_obj.assign(nr, -1);
    R.rec_shape.assign(nr, -1);
    for (const GRec& r : all) {
        int64_t dst = (int64_t)R.first_rec[rank[r.node]] + r.k;
        const RayS* rs[3] = {&r.g.c, &r.g.w, &r.g.d};
        const Hit* hs[3] = {&r.o.Xc, &r.o.Xw, &r.o.Xd};
        for (int b = 0; b < 3; ++b) {
            double vals[11] = {rs[b]->pos.x, rs[b]->pos.y, rs[b]->pos.z, rs[b]->dir.x, rs[b]->dir.y, rs[b]->dir.z, rs[b]->n,
                               hs[b]->t, hs[b]->n.x, hs[b]->n.y, hs[b]->n.z};
            for (int p2 = 0; p2 < 11; ++p2) R.rec[(size_t)(11 * b + p2) * nr + dst] = vals[p2];
        }
        R.rec_obj[dst] = r.o.Xc.obj;
        R.rec_shape[dst] = r.o.Xc.shape;
    }
    const int ndet = d->n_detectors;
    R.det_count.assign(ndet, 0);
    R.det_offset.assign(ndet, 0);
    for (int q = 0; q < ndet; ++q) {
        R.det_offset[q] = (int64_t)R.det_node.size();
        for (int64_t i = 0; i < nn; ++i) {
            const GNode& nd = nodes[order[i]];
            if (nd.hit_det != q) continue;
            for (int sub = 0; sub < 3; ++sub) {
                R.det_node.push_back((int32_t)i);
                R.det.insert(R.det.end(), nd.hit + 9 * sub, nd.hit + 9 * sub + 9);
            }
        }
        R.det_count[q] = (int64_t)R.det_node.size() - R.det_offset[q];
    }
    std::memset(v, 0, sizeof *v);
    v->n_roots = n;
    v->n_nodes = nn;
    v->n_records = nr;
    v->n_intersect_calls = (int64_t)calls;
    v->n_steps = steps;
    v->beam_kind = BMO_BEAM_GAUSSIAN;
    v->rec_planes = PL;
    v->n_detectors = ndet;
    v->node_root = R.root.data();
    v->node_parent = R.parent.data();
    v->node_first_child = R.first_child.data();
    v->node_first_rec = R.first_rec.data();
    v->node_nseg = R.nseg.data();
    v->node_status = R.status.data();
    v->node_aux = R.aux.data();
    v->rec_obj = R.rec_obj.data();
    v->rec_shape = R.rec_shape.data();
    v->rec = R.rec.data();
    v->det_count = R.det_count.data();
    v->det_offset = R.det_offset.data();
    v->det_node = R.det_node.data();
    v->det_data = R.det.data();
}
}  // namespace

extern "C" {
// the lane code's elementary functions one by one (bmo_jlmath.hpp; tests/test_jl_trig.py): 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 atan(y, x)
void bmo_emu_jl_trig_n(int which, const double* x, const double* y, long long n, double* out) {
    for (long long i = 0; i < n; ++i) {
        const double a = x[i], b = y ? y[i] : 0.0;
        out[i] = which == 0 ? bmo::jl::sin(a) : which == 1 ? bmo::jl::cos(a) : which == 2 ? bmo::jl::tan(a) : which == 3 ? bmo::jl::acos(a)
               : which == 4 ? bmo::jl::atan(a) : bmo::jl::atan2(b, a);
    }
}
int bmo_emu_trace(const bmo_scene_desc* d, const bmo_ray_batch* in, const bmo_trace_opts* opts, void** handle, bmo_trace_result_view* v) {
    auto* R = new ResultE();
    try {
        if (in->kind == BMO_BEAM_RAY) run<BMO_BEAM_RAY>(d, in, opts, *R, v);
        else if (in->kind == BMO_BEAM_POLARIZED) run<BMO_BEAM_POLARIZED>(d, in, opts, *R, v);
        else if (in->kind == BMO_BEAM_GAUSSIAN) run_gauss(d, in, opts, *R, v);
        else {
            delete R;
            return BMO_ERR_UNSUPPORTED;
        }
    } catch (const BeamLimit&) {
        delete R;
        return BMO_ERR_LIMIT;
    }
    *handle = R;
    return BMO_OK;
}
// second solve of already solved beams: `prev` is the canonical view of the previous solution (any backend's)
int bmo_emu_retrace(const bmo_scene_desc* d, const bmo_ray_batch* in, const bmo_trace_opts* opts, const bmo_trace_result_view* prev, void** handle,
                    bmo_trace_result_view* v) {
    if (!prev || prev->n_roots != in->n || prev->beam_kind != in->kind) return BMO_ERR_INVALID;
    auto* R = new ResultE();
    try {
        if (in->kind == BMO_BEAM_RAY) run<BMO_BEAM_RAY>(d, in, opts, *R, v, prev);
        else if (in->kind == BMO_BEAM_POLARIZED) run<BMO_BEAM_POLARIZED>(d, in, opts, *R, v, prev);
        else if (in->kind == BMO_BEAM_GAUSSIAN) run_gauss(d, in, opts, *R, v, prev);
        else {
            delete R;
            return BMO_ERR_UNSUPPORTED;
        }
    } catch (const BeamLimit&) {
        delete R;
        return BMO_ERR_LIMIT;
    }
    *handle = R;
    return BMO_OK;
}
int bmo_emu_free(void* h) {
    delete static_cast<ResultE*>(h);
    return 0;
}
}

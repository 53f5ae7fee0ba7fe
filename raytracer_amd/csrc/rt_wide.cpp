// Wide trees for the opt-in RT_TRAVERSAL_WIDE walk (no counterpart in the reference, whose trees
// are binary: Bvh shape.rs:87-139, MeshBvhNode shape.rs:185-192).
//
// The flattened pre-order entry array of a scene (either build mode) is collapsed into nodes of up
// to eight children: starting from a box's children, the child box of largest surface area is
// replaced by its own children until eight are reached.  A node stores its children's boxes itself,
// quantised to 8 bits per plane on a power-of-two grid local to the node (80 bytes for eight boxes
// where the binary tree reads 32 bytes per box), so one visit fetches and tests what took the
// binary walk up to three dependent round trips.  Primitives are the leaves, one per child slot,
// each with a box of its own (the binary trees test a triangle without one).
//
// Exactness: the primitive tests are the reference's (tri_step / sphere_step on the same entry
// words), and the closest hit is a minimum over the primitives whose boxes the ray passes.  A
// quantised box encloses the true box with at least one grid step to spare on every side, and the
// grid step is at least four ulps of the node's largest coordinate, so the slab test on it, done
// in a fused form (one FMA per plane), passes whenever the reference's test on the true box does
// -- up to the same caveat as for the other opt-in trees: a primitive whose own test accepts a
// point that rounding puts outside its padded box would be lost, and ties in t may resolve to
// another primitive.
//
// Rays whose direction is so long or so short that the sphere test leaves the range of f32: the
// reference accepts a sphere whose discriminant overflowed to NaN (shape.rs:25-36) wherever it
// tests one, with |d|^2 a denormal its root lies anywhere, and it tests the objects of a
// HittableList at the root without any box before them.  Word 7 of a node marks the children that
// stand for such objects; when a ray's |d|^2 is outside [1e-30, 1e18) (or NaN) the walk enters
// them whatever their boxes say, as the reference does.  Other rays keep the culling.  (Primitives
// under boxes stay behind their own boxes for such rays too: there the reference's result depends
// on which of its boxes rounding leaves open, which no other tree can repeat.)
//
// Transformed meshes (Mesh::hit transforms the ray before it descends, shape.rs:454): the tree of such
// a mesh -- the entries between its ENTER and LEAVE markers -- is collapsed on its own, in the mesh's
// space, and its root node carries the transform's number in the upper bits of word 7.  In the node
// above it is an ordinary node child whose box is the mesh's local root box taken to the caller's
// space (its eight corners, padded).  The walk transforms a lane's ray when it fetches such a root and
// restores it when the subtree is exhausted; t is the same parameter in both spaces.
//
// Child slots and order: a child sits in the slot whose diagonal (bit k of the slot index: the
// positive side along axis k) matches the offset of its centre from the node's centre best, so a
// ray whose direction is positive along the axes in `r` meets the children roughly in descending
// order of (slot ^ r) -- the stack walk takes them in that order without sorting distances.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <exception>
#include <memory>
#include <string>
#include <thread>

#include "rt_internal.hpp"

namespace rtamd {
namespace {

struct WBox { float lo[3], hi[3]; };

inline WBox wempty() {
    const float inf = std::numeric_limits<float>::infinity();
    return WBox{{inf, inf, inf}, {-inf, -inf, -inf}};
}
inline bool wvalid(const WBox& b) { return b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]; }
inline void wgrow(WBox& a, const WBox& b) {
    for (int k = 0; k < 3; k++) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
}
inline double warea(const WBox& b) {
    if (!wvalid(b)) return 0.0;
    const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
    const double a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : std::numeric_limits<double>::max();    // (inf - inf)
}

struct Arena {
    std::vector<uint32_t> nodes;   // WIDE_NODE_WORDS per node slot; slot 0 is the arena's root
    std::vector<float> prims;      // WIDE_PRIM_FLOATS per primitive
    uint32_t depth = 0;            // deepest level that holds a node (the root's level is 1)
    std::string error;
};

// one candidate child of a node under construction: an entry of the flattened array, or (only
// where a container holds more than eight objects) a group of candidates made up here
struct Item {
    size_t entry = 0;
    std::shared_ptr<std::vector<Item>> group;
    WBox box;
    size_t span = 1;               // entries beneath (what decides whether a subtree gets a thread)
    uint32_t xform = 0;            // a transformed mesh (entry = its ENTER marker): the transform's index + 1
    // a primitive the reference tests without any box test before it (an object of a HittableList at
    // the root, shape.rs:60-74, or the root itself), or a group made up here that holds one
    bool always = false;
};

struct WideBuilder {
    const float* E;
    size_t n;
    const float* X = nullptr;      // 32 floats per transform: its four columns, then its inverse's

    uint32_t word(size_t i, int w) const { uint32_t u; std::memcpy(&u, E + i * ENTRY_FLOATS + w, 4); return u; }
    uint32_t tag(size_t i) const { return word(i, 3) >> TAG_SHIFT; }
    size_t skip(size_t i) const { return word(i, 3) & SKIP_MASK; }
    bool is_enter(size_t i) const { return tag(i) == TAG_END && word(i, 0) == END_CODE_ENTER; }
    size_t next_sibling(size_t i) const {
        if (is_enter(i)) return next_sibling(i + 1) + 1;       // ENTER, the mesh's tree, LEAVE
        return tag(i) == TAG_BOX ? skip(i) : i + 1;
    }

    // A primitive's own box: the triangle the intersection routine sees (v0, v0 + e1, v0 + e2) or the
    // sphere, grown by the reference's 0.0001 (compute_aabb, shape.rs:225-231) plus four ulps of its
    // largest coordinate (the corners are recomputed from rounded edges here).
    WBox prim_box(size_t i) const {
        const float* e = E + i * ENTRY_FLOATS;
        WBox b;
        float big = 0.0f;
        if (tag(i) == TAG_TRI) {
            for (int k = 0; k < 3; k++) {
                const float p0 = e[k], p1 = e[k] + e[4 + k], p2 = e[k] + e[8 + k];
                b.lo[k] = std::min(p0, std::min(p1, p2));
                b.hi[k] = std::max(p0, std::max(p1, p2));
            }
        } else {
            const float r = std::fabs(e[4]);
            for (int k = 0; k < 3; k++) { b.lo[k] = e[k] - r; b.hi[k] = e[k] + r; }
        }
        for (int k = 0; k < 3; k++) big = std::max(big, std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
        const float pad = 0.0001f + big * 4.76837158e-7f;      // 2^-21
        for (int k = 0; k < 3; k++) { b.lo[k] -= pad; b.hi[k] += pad; }
        return b;
    }
    WBox stored_box(size_t i) const {
        const float* e = E + i * ENTRY_FLOATS;
        return WBox{{e[0], e[1], e[2]}, {e[4], e[5], e[6]}};
    }
    Item entry_item(size_t i) const {
        Item it;
        it.entry = i;
        if (is_enter(i)) {
            // the mesh's local root box in the caller's space: its corners through the transform, padded by
            // what the rounding of the transformed ray and of the corners may cost
            const Item root = entry_item(i + 1);
            it.xform = word(i, 1) + 1u;
            it.span = root.span + 2;
            it.box = wempty();
            const float* m = X + (size_t)(it.xform - 1u) * 32;
            for (int c = 0; c < 8; c++) {
                const double x = (c & 1) ? root.box.hi[0] : root.box.lo[0], y = (c & 2) ? root.box.hi[1] : root.box.lo[1],
                             z = (c & 4) ? root.box.hi[2] : root.box.lo[2];
                for (int k = 0; k < 3; k++) {
                    const double v = (double)m[k] * x + (double)m[4 + k] * y + (double)m[8 + k] * z + (double)m[12 + k];
                    it.box.lo[k] = std::min(it.box.lo[k], (float)std::nextafter((float)v, -std::numeric_limits<float>::infinity()));
                    it.box.hi[k] = std::max(it.box.hi[k], (float)std::nextafter((float)v, std::numeric_limits<float>::infinity()));
                }
            }
            float big = 0.0f;
            for (int k = 0; k < 3; k++) big = std::max(big, std::max(std::fabs(it.box.lo[k]), std::fabs(it.box.hi[k])));
            const float pad = 0.0002f + big * 1.52587891e-5f;     // 2^-16
            for (int k = 0; k < 3; k++) { it.box.lo[k] -= pad; it.box.hi[k] += pad; }
            return it;
        }
        if (tag(i) == TAG_BOX) { it.box = stored_box(i); it.span = skip(i) - i; }
        else { it.box = prim_box(i); it.span = 1; }
        return it;
    }
    bool is_prim(const Item& it) const { return !it.group && !it.xform && tag(it.entry) != TAG_BOX; }
    size_t child_count(const Item& it) const {
        if (it.group) return it.group->size();
        if (it.xform) return 1;
        size_t k = 0;
        for (size_t c = it.entry + 1, end = skip(it.entry); c < end; c = next_sibling(c)) k++;
        return k;
    }
    void children(const Item& it, std::vector<Item>& out) const {
        if (it.group) { out.insert(out.end(), it.group->begin(), it.group->end()); return; }
        if (it.xform) { out.push_back(entry_item(it.entry + 1)); return; }     // the mesh's own root, in its own space
        for (size_t c = it.entry + 1, end = skip(it.entry); c < end; c = next_sibling(c)) out.push_back(entry_item(c));
    }

    // more than eight candidates that cannot be merged by the trees themselves (a HittableList of many
    // objects, a leaf of many triangles): consecutive runs along the axis their centres spread most on
    void chunk(std::vector<Item>& items) const {
        WBox cb = wempty();
        for (const Item& it : items)
            for (int k = 0; k < 3; k++) {
                const float c = 0.5f * it.box.lo[k] + 0.5f * it.box.hi[k];
                if (c == c) { cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c); }
            }
        int axis = 0;
        for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        std::stable_sort(items.begin(), items.end(), [axis](const Item& a, const Item& b) {
            return a.box.lo[axis] + a.box.hi[axis] < b.box.lo[axis] + b.box.hi[axis];
        });
        const size_t total = items.size(), per = (total + 7) / 8;
        std::vector<Item> out;
        for (size_t i = 0; i < total; i += per) {
            const size_t m = std::min(per, total - i);
            if (m == 1) { out.push_back(items[i]); continue; }
            Item g;
            g.group = std::make_shared<std::vector<Item>>(items.begin() + i, items.begin() + i + m);
            g.box = wempty();
            g.span = 0;
            for (const Item& it : *g.group) { wgrow(g.box, it.box); g.span += it.span; g.always = g.always || it.always; }
            out.push_back(std::move(g));
        }
        items.swap(out);
    }

    // grid of one axis of a node: step 2^(e - 127) and origin such that every child plane lies between
    // grid points 1 and 254 with a step to spare
    static bool grid_for(float lo, float hi, float& origin, uint32_t& e) {
        if (!(lo <= hi) || std::isinf(lo) || std::isinf(hi)) return false;
        const double ext = (double)hi - (double)lo;
        const double big = std::max(std::fabs((double)lo), std::fabs((double)hi));
        int k = -120;
        if (ext > 0.0) k = std::max(k, (int)std::ceil(std::log2(ext / 250.0)));
        if (big > 0.0) k = std::max(k, std::ilogb(big) - 23 + 2);      // >= 4 ulps of the largest coordinate
        for (;; k++) {
            if (k > 126) return false;
            const double s = std::ldexp(1.0, k);
            const float o = (float)((double)lo - 2.0 * s);
            if (std::isinf(o)) return false;
            // planes: floor((lo - o) / s) - 1 >= 0 and ceil((hi - o) / s) + 1 <= 255
            if (std::floor(((double)lo - o) / s) - 1.0 >= 0.0 && std::ceil(((double)hi - o) / s) + 1.0 <= 255.0) {
                origin = o;
                e = (uint32_t)(k + 127);
                return e >= 1 && e <= 254;
            }
        }
    }

    // Builds the node whose children are drawn from `items` at slot `at` of the arena (its descendants
    // are appended) and returns the box of what it holds.
    WBox make_node(std::vector<Item> items, Arena& A, size_t at, uint32_t depth, int fork_levels) const {
        if (!A.error.empty()) return wempty();
        if (depth > WIDE_MAX_DEPTH) { A.error = "the wide tree is deeper than its walk's stack"; return wempty(); }
        // collapse: open the largest box while the children still fit eight slots
        std::vector<Item> kids;
        for (;;) {
            long best = -1;
            double best_area = -1.0;
            bool dropped = false;
            for (size_t i = 0; i < items.size(); i++) {
                if (is_prim(items[i]) || items[i].xform) continue;     // (a transformed mesh stays a node of its own)
                const size_t k = child_count(items[i]);
                if (k == 0) { items.erase(items.begin() + (long)i); dropped = true; break; }
                if (items.size() - 1 + k > 8) continue;
                const double a = k == 1 ? std::numeric_limits<double>::infinity() : warea(items[i].box);
                if (a > best_area) { best_area = a; best = (long)i; }
            }
            if (dropped) continue;
            if (best < 0) break;
            kids.clear();
            children(items[(size_t)best], kids);
            items.erase(items.begin() + best);
            items.insert(items.end(), kids.begin(), kids.end());
        }
        while (items.size() > 8) chunk(items);
        A.depth = std::max(A.depth, depth);

        const size_t nc = items.size();
        // slots: greedily the (child, slot) pair whose diagonal fits the child's offset best
        int slot_of[8];
        {
            WBox all = wempty();
            for (const Item& it : items) wgrow(all, it.box);
            double score[8][8];
            for (size_t c = 0; c < nc; c++)
                for (int s = 0; s < 8; s++) {
                    double v = 0.0;
                    for (int k = 0; k < 3; k++) {
                        double d = ((double)items[c].box.lo[k] + items[c].box.hi[k]) - ((double)all.lo[k] + all.hi[k]);
                        if (!(d == d)) d = 0.0;
                        v += ((s >> k) & 1) ? d : -d;
                    }
                    score[c][s] = v;
                }
            bool child_done[8] = {}, slot_done[8] = {};
            for (size_t round = 0; round < nc; round++) {
                int bc = -1, bs = -1;
                for (size_t c = 0; c < nc; c++) {
                    if (child_done[c]) continue;
                    for (int s = 0; s < 8; s++)
                        if (!slot_done[s] && (bc < 0 || score[c][s] > score[bc][bs])) { bc = (int)c; bs = s; }
                }
                child_done[bc] = true; slot_done[bs] = true; slot_of[bc] = bs;
            }
        }
        int child_in[8];
        std::fill(child_in, child_in + 8, -1);
        for (size_t c = 0; c < nc; c++) child_in[slot_of[c]] = (int)c;

        uint32_t imask = 0, lmask = 0, amask = 0;
        for (int s = 0; s < 8; s++)
            if (child_in[s] >= 0) {
                (is_prim(items[(size_t)child_in[s]]) ? lmask : imask) |= 1u << s;
                if (items[(size_t)child_in[s]].always) amask |= 1u << s;
            }
        size_t child_base = 0;
        if (imask) {
            child_base = A.nodes.size() / WIDE_NODE_WORDS;
            A.nodes.resize(A.nodes.size() + 8 * WIDE_NODE_WORDS, 0u);
        }
        const size_t prim_base = A.prims.size() / WIDE_PRIM_FLOATS;
        WBox cbox[8];
        for (int s = 0; s < 8; s++) {
            cbox[s] = wempty();
            if (!(lmask >> s & 1)) continue;
            const Item& it = items[(size_t)child_in[s]];
            const size_t at_p = A.prims.size();
            A.prims.resize(at_p + WIDE_PRIM_FLOATS, 0.0f);
            std::memcpy(&A.prims[at_p], E + it.entry * ENTRY_FLOATS, 12 * sizeof(float));
            cbox[s] = it.box;
        }
        // internal children: big subtrees near the root on threads of their own, each into its own
        // arena, appended afterwards in slot order (the layout does not depend on timing)
        std::unique_ptr<Arena> sub[8];
        std::thread workers[8];
        struct JoinAll {                           // whatever happens below, no thread is left running
            std::thread* t;
            ~JoinAll() { for (int s = 0; s < 8; s++) if (t[s].joinable()) t[s].join(); }
        } join_all{workers};
        for (int s = 0; s < 8; s++) {
            if (!(imask >> s & 1)) continue;
            const Item& it = items[(size_t)child_in[s]];
            if (fork_levels > 0 && it.span >= (1u << 16)) {
                sub[s].reset(new Arena());
                sub[s]->nodes.assign(WIDE_NODE_WORDS, 0u);
                Arena* B = sub[s].get();
                WBox* out = &cbox[s];
                std::vector<Item> ck;
                children(it, ck);
                workers[s] = std::thread([this, ck, B, out, depth, fork_levels]() {
                    try {                              // (nothing may leave a thread: an exception there ends the process)
                        *out = make_node(ck, *B, 0, depth + 1, fork_levels - 1);
                    } catch (const std::exception& e) {
                        B->error = std::string("wide tree: ") + e.what();
                    } catch (...) {
                        B->error = "wide tree: unknown failure";
                    }
                });
            }
        }
        for (int s = 0; s < 8; s++) {
            if (!(imask >> s & 1) || sub[s]) continue;
            std::vector<Item> ck;
            children(items[(size_t)child_in[s]], ck);
            cbox[s] = make_node(std::move(ck), A, child_base + (size_t)s, depth + 1, 0);
        }
        for (int s = 0; s < 8; s++) {
            if (!sub[s]) continue;
            workers[s].join();
            Arena& B = *sub[s];
            if (!B.error.empty()) { if (A.error.empty()) A.error = B.error; continue; }
            const uint32_t node_off = (uint32_t)(A.nodes.size() / WIDE_NODE_WORDS) - 1u;
            const uint32_t prim_off = (uint32_t)(A.prims.size() / WIDE_PRIM_FLOATS);
            for (size_t i = 0; i < B.nodes.size(); i += WIDE_NODE_WORDS) {
                if (B.nodes[i + 3] >> 24) B.nodes[i + 4] += node_off;
                B.nodes[i + 5] += prim_off;
            }
            std::memcpy(&A.nodes[(child_base + (size_t)s) * WIDE_NODE_WORDS], B.nodes.data(), WIDE_NODE_WORDS * 4);
            A.nodes.insert(A.nodes.end(), B.nodes.begin() + WIDE_NODE_WORDS, B.nodes.end());
            A.prims.insert(A.prims.end(), B.prims.begin(), B.prims.end());
            A.depth = std::max(A.depth, B.depth);
            sub[s].reset();
        }
        if (!A.error.empty()) return wempty();

        // a child that turned out empty (a container of nothing) leaves its slot
        for (int s = 0; s < 8; s++)
            if ((imask >> s & 1) && !wvalid(cbox[s])) imask &= ~(1u << s);
        // a transformed mesh: its root node names the transform, and its box up here is the one in this space
        for (int s = 0; s < 8; s++) {
            if (!(imask >> s & 1)) continue;
            const Item& it = items[(size_t)child_in[s]];
            if (!it.xform) continue;
            A.nodes[(child_base + (size_t)s) * WIDE_NODE_WORDS + 7] |= it.xform << 8;
            cbox[s] = it.box;
        }
        WBox all = wempty();
        for (int s = 0; s < 8; s++)
            if ((imask | lmask) >> s & 1) wgrow(all, cbox[s]);

        uint32_t w[WIDE_NODE_WORDS] = {};
        if (imask | lmask) {
            float origin[3];
            uint32_t e[3];
            for (int k = 0; k < 3; k++)
                if (!grid_for(all.lo[k], all.hi[k], origin[k], e[k])) {
                    A.error = "wide tree: a box is not finite";
                    return wempty();
                }
            uint8_t q[6][8] = {};
            for (int s = 0; s < 8; s++) {
                if (!((imask | lmask) >> s & 1)) continue;
                for (int k = 0; k < 3; k++) {
                    const double st = std::ldexp(1.0, (int)e[k] - 127);
                    const double ql = std::floor(((double)cbox[s].lo[k] - origin[k]) / st) - 1.0;
                    const double qh = std::ceil(((double)cbox[s].hi[k] - origin[k]) / st) + 1.0;
                    if (ql < 0.0 || qh > 255.0 || !(origin[k] + ql * st <= (double)cbox[s].lo[k] - 0.5 * st) ||
                        !(origin[k] + qh * st >= (double)cbox[s].hi[k] + 0.5 * st)) {
                        A.error = "internal error: wide tree quantisation";
                        return wempty();
                    }
                    q[k][s] = (uint8_t)ql;
                    q[3 + k][s] = (uint8_t)qh;
                }
            }
            for (int k = 0; k < 3; k++) std::memcpy(&w[k], &origin[k], 4);
            w[3] = e[0] | e[1] << 8 | e[2] << 16 | imask << 24;
            for (int p = 0; p < 6; p++) {
                w[8 + 2 * p] = q[p][0] | q[p][1] << 8 | q[p][2] << 16 | (uint32_t)q[p][3] << 24;
                w[9 + 2 * p] = q[p][4] | q[p][5] << 8 | q[p][6] << 16 | (uint32_t)q[p][7] << 24;
            }
        }
        w[4] = imask ? (uint32_t)child_base : 0u;
        w[5] = (uint32_t)prim_base;
        w[6] = lmask;
        w[7] = amask & (imask | lmask);
        std::memcpy(&A.nodes[at * WIDE_NODE_WORDS], w, sizeof(w));
        return all;
    }
};

}  // namespace

// The builder hands every node with node children a block of eight slots, one per child slot, and
// most blocks stay half empty.  Here the nodes move to new slots so that the blocks overlap: a
// node's children still live at [4] + child slot, but the base is the lowest one at which the slots
// its children need are free (first fit, in depth-first order, so a family stays within a few
// slots and near its parent), and the holes of one family are taken by the next.  The walk does not
// change; the array shrinks to little more than the nodes there are.
void compact_nodes(std::vector<uint32_t>& nodes) {
    const size_t n_old = nodes.size() / WIDE_NODE_WORDS;
    if (n_old <= 1) return;
    std::vector<uint32_t> out(WIDE_NODE_WORDS, 0u);
    std::vector<uint8_t> used(1, 1);           // slot 0: the root
    size_t low = 1;                            // every slot below is taken
    std::memcpy(out.data(), nodes.data(), WIDE_NODE_WORDS * 4);
    std::vector<uint32_t> stack{0u};           // new slots of nodes whose children are still to be placed
    while (!stack.empty()) {
        const uint32_t at = stack.back();
        stack.pop_back();
        const uint32_t imask = out[(size_t)at * WIDE_NODE_WORDS + 3] >> 24;
        if (!imask) continue;
        const uint32_t old_base = out[(size_t)at * WIDE_NODE_WORDS + 4];
        const int first = __builtin_ctz(imask);
        // (holes more than a few blocks behind the highest slot in use are given up: the search stays short)
        const size_t from = std::max(low, used.size() > 64 ? used.size() - 64 : 0);
        size_t base = from > (size_t)first ? from - (size_t)first : 0;
        for (;; base++) {
            bool fits = true;
            for (int s = first; s < 8 && fits; s++)
                if ((imask >> s & 1) && base + (size_t)s < used.size() && used[base + (size_t)s]) fits = false;
            if (fits) break;
        }
        if (base + 8 > used.size()) { used.resize(base + 8, 0); out.resize((base + 8) * WIDE_NODE_WORDS, 0u); }
        out[(size_t)at * WIDE_NODE_WORDS + 4] = (uint32_t)base;
        for (int s = 7; s >= 0; s--) {         // (pushed in reverse: the child in the lowest slot is placed first)
            if (!(imask >> s & 1)) continue;
            used[base + (size_t)s] = 1;
            std::memcpy(&out[(base + (size_t)s) * WIDE_NODE_WORDS], &nodes[((size_t)old_base + (size_t)s) * WIDE_NODE_WORDS],
                        WIDE_NODE_WORDS * 4);
            stack.push_back((uint32_t)(base + (size_t)s));
        }
        while (low < used.size() && used[low]) low++;
    }
    size_t last = used.size();
    while (last > 1 && !used[last - 1]) last--;
    out.resize(last * WIDE_NODE_WORDS);
    nodes.swap(out);
}

std::string build_wide_scene(const float* entries, size_t n_entries, WideScene& out, const float* xforms, size_t n_xforms) {
    out = WideScene();
    WideBuilder b{entries, n_entries, xforms};
    std::vector<Item> top;
    for (size_t i = 0; i < n_entries; i = b.next_sibling(i)) {
        if (b.tag(i) == TAG_END && !(b.is_enter(i) && xforms && b.word(i, 1) < n_xforms)) return "internal error: wide tree: a stray marker";
        top.push_back(b.entry_item(i));
        top.back().always = b.tag(i) != TAG_BOX && !top.back().xform;
    }
    Arena A;
    A.nodes.assign(WIDE_NODE_WORDS, 0u);
    b.make_node(std::move(top), A, 0, 1, 2);
    if (!A.error.empty()) return A.error;
    if (!std::getenv("RT_WIDE_SPARSE")) compact_nodes(A.nodes);        // (experiments: the blocks as the builder made them)
    if (A.nodes.size() / WIDE_NODE_WORDS >= (1u << 24)) return "the wide tree has more than 2^24 node slots";
    const char* force = std::getenv("RT_WIDE_STRIDE");                  // (experiments: 20 or 32 whatever the size)
    const bool lines = force ? std::atoi(force) == (int)WIDE_LINE_WORDS : A.nodes.size() * 4 > WIDE_PACKED_MAX_BYTES;
    if (lines) {                               // one node per 128-byte line
        const size_t n_slots = A.nodes.size() / WIDE_NODE_WORDS;
        std::vector<uint32_t> wide(n_slots * WIDE_LINE_WORDS, 0u);
        for (size_t i = 0; i < n_slots; i++)
            std::memcpy(&wide[i * WIDE_LINE_WORDS], &A.nodes[i * WIDE_NODE_WORDS], WIDE_NODE_WORDS * 4);
        A.nodes.swap(wide);
        out.node_words = WIDE_LINE_WORDS;
    }
    out.nodes.swap(A.nodes);
    out.prims.swap(A.prims);
    out.depth = std::max(1u, A.depth);
    return std::string();
}

}  // namespace rtamd

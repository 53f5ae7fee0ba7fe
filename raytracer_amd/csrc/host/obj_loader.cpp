// Wavefront OBJ / MTL reader with the semantics Mesh::from_file relies on (shape.rs:352-363:
// tobj::load_obj with single_index + triangulate + ignore_points + ignore_lines):
//   * a new model starts at every `o` / `g`, and at a `usemtl` that changes the material while
//     faces are pending;
//   * polygons are fanned from their first vertex: a b c d -> (a,b,c), (a,c,d);
//   * a model's material id is the index of its `usemtl` name in the MTL file's newmtl order;
//   * `mtllib` is resolved relative to the OBJ file's directory.
// Only positions and normals are kept (texture coordinates are never read by the renderer).
//
// Built for big files (a 10 M-triangle scene is 1.5 GB of text): the file is cut at line ends into
// one chunk per thread.  Pass 1 counts the `v` / `vn` lines of every chunk, so that each chunk
// knows how many vertices precede it (relative indices and range checks need that).  Pass 2 parses
// the chunks in parallel -- numbers into their final places, faces into chunk-local index lists --
// and only notes the lines that change state (`o`, `g`, `usemtl`, `mtllib`).  A serial merge then
// replays those notes in file order and cuts the face lists into models, exactly as a single pass
// over the file would.  Small files take the same code with one chunk.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include "scene_api.hpp"

namespace rthost {
namespace {

bool read_file(const std::string& path, std::string& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? std::fread(&out[0], 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == out.size();
}

inline const char* skip_ws(const char* p) { while (*p == ' ' || *p == '\t') p++; return p; }
inline bool is_eol(char c) { return c == '\n' || c == '\r' || c == '\0'; }

std::string rest_of_line(const char* p) {
    p = skip_ws(p);
    const char* e = p;
    while (!is_eol(*e)) e++;
    while (e > p && (e[-1] == ' ' || e[-1] == '\t')) e--;
    return std::string(p, e);
}
std::string first_token(const char* p) {
    p = skip_ws(p);
    const char* e = p;
    while (!is_eol(*e) && *e != ' ' && *e != '\t') e++;
    return std::string(p, e);
}

void load_mtl_names(const std::string& path, ObjData& out, std::map<std::string, int>& mat_map) {
    std::string text;
    if (!read_file(path, text)) return;
    const char* p = text.c_str();
    while (*p) {
        const char* line = skip_ws(p);
        if (std::strncmp(line, "newmtl", 6) == 0 && (line[6] == ' ' || line[6] == '\t')) {
            std::string name = first_token(line + 6);
            mat_map[name] = (int)out.material_names.size();
            out.material_names.push_back(name);
        }
        while (*p && *p != '\n') p++;
        if (*p) p++;
    }
}

}  // namespace

namespace {

struct ObjEvent { char kind; size_t at; std::string arg; };   // kind: 'o' (o / g), 'u' (usemtl), 'm' (mtllib); at: corners before it

struct ObjChunk {
    const char* begin = nullptr;
    const char* end = nullptr;
    size_t n_v = 0, n_vn = 0;                  // lines of this chunk
    size_t v0 = 0, vn0 = 0;                    // vertices / normals in the chunks before it
    std::vector<uint32_t> vi, ni;              // triangulated corners
    std::vector<ObjEvent> events;
    size_t n_faces = 0;
    std::string error;
};

inline const char* next_line(const char* p, const char* end) {
    while (p < end && *p != '\n') p++;
    return p < end ? p + 1 : end;
}

void count_chunk(ObjChunk& c) {
    for (const char* p = c.begin; p < c.end; p = next_line(p, c.end)) {
        const char* q = skip_ws(p);
        if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) c.n_v++;
        else if (q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) c.n_vn++;
    }
}

void parse_chunk(ObjChunk& c, float* positions, float* normals, const std::string& path) {
    size_t nv = c.v0, nn = c.vn0;              // defined so far, as a single pass would have them
    std::vector<long> fv, fn;
    for (const char* p = c.begin; p < c.end; p = next_line(p, c.end)) {
        const char* q = skip_ws(p);
        if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
            char* e;
            q += 1;
            for (int k = 0; k < 3; k++) { positions[nv * 3 + k] = std::strtof(q, &e); q = e; }
            nv++;
        } else if (q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) {
            char* e;
            q += 2;
            for (int k = 0; k < 3; k++) { normals[nn * 3 + k] = std::strtof(q, &e); q = e; }
            nn++;
        } else if (q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
            fv.clear(); fn.clear();
            q += 1;
            for (;;) {
                q = skip_ws(q);
                if (is_eol(*q)) break;
                char* e;
                long v = std::strtol(q, &e, 10);
                if (e == q) { c.error = "bad face in " + path; return; }
                long n = 0; bool have_n = false;
                q = e;
                if (*q == '/') {
                    q++;
                    if (*q != '/') { (void)std::strtol(q, &e, 10); q = e; }    // vt, unused
                    if (*q == '/') {
                        q++;
                        n = std::strtol(q, &e, 10);
                        have_n = e != q;
                        q = e;
                    }
                }
                fv.push_back(v > 0 ? v - 1 : (long)nv + v);
                fn.push_back(have_n ? (n > 0 ? n - 1 : (long)nn + n) : -1);
            }
            if (fv.size() >= 3) {              // points and lines are ignored
                c.n_faces++;
                for (size_t i = 1; i + 1 < fv.size(); i++) {
                    const size_t corner[3] = {0, i, i + 1};
                    for (size_t k : corner) {
                        if (fv[k] < 0 || fv[k] >= (long)nv) { c.error = "face index out of range in " + path; return; }
                        if (fn[k] < 0 || fn[k] >= (long)nn) { c.error = "face without a valid normal in " + path; return; }
                        c.vi.push_back((uint32_t)fv[k]);
                        c.ni.push_back((uint32_t)fn[k]);
                    }
                }
            }
        } else if ((q[0] == 'o' || q[0] == 'g') && (q[1] == ' ' || q[1] == '\t' || is_eol(q[1]))) {
            c.events.push_back(ObjEvent{'o', c.vi.size(), rest_of_line(q + 1)});
        } else if (std::strncmp(q, "usemtl", 6) == 0 && (q[6] == ' ' || q[6] == '\t')) {
            c.events.push_back(ObjEvent{'u', c.vi.size(), first_token(q + 6)});
        } else if (std::strncmp(q, "mtllib", 6) == 0 && (q[6] == ' ' || q[6] == '\t')) {
            c.events.push_back(ObjEvent{'m', c.vi.size(), first_token(q + 6)});
        }
    }
}

}  // namespace

std::string load_obj(const std::string& path, ObjData& out, int threads) {
    out = ObjData();
    std::string text;
    if (!read_file(path, text)) return "cannot read " + path;
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash + 1);

    // chunks: one per thread above 4 MB each (RT_OBJ_THREADS forces a count, for tests)
    if (threads <= 0) {
        if (const char* e = std::getenv("RT_OBJ_THREADS")) threads = std::atoi(e);
    }
    if (threads <= 0) {
        const size_t hw = std::max(1u, std::thread::hardware_concurrency());
        threads = (int)std::min<size_t>(std::min<size_t>(hw, 32), std::max<size_t>(1, text.size() / ((size_t)4 << 20)));
    }
    const char* const t0 = text.c_str();
    const char* const t1 = t0 + text.size();
    std::vector<ObjChunk> chunks((size_t)threads);
    const char* cut = t0;
    for (int k = 0; k < threads; k++) {
        chunks[k].begin = cut;
        const char* want = k + 1 == threads ? t1 : t0 + text.size() * (size_t)(k + 1) / (size_t)threads;
        if (want < cut) want = cut;
        cut = k + 1 == threads ? t1 : next_line(want, t1);     // to the end of the line `want` falls into
        chunks[k].end = cut;
    }
    auto for_chunks = [&](auto&& fn) {
        if (threads == 1) { fn(chunks[0]); return; }
        std::vector<std::thread> pool;
        for (ObjChunk& c : chunks) pool.emplace_back([&fn, &c]() { fn(c); });
        for (std::thread& t : pool) t.join();
    };
    for_chunks([](ObjChunk& c) { count_chunk(c); });
    size_t nv = 0, nn = 0;
    for (ObjChunk& c : chunks) { c.v0 = nv; c.vn0 = nn; nv += c.n_v; nn += c.n_vn; }
    out.positions.resize(nv * 3);
    out.normals.resize(nn * 3);
    float* positions = out.positions.data();
    float* normals = out.normals.data();
    for_chunks([&](ObjChunk& c) { parse_chunk(c, positions, normals, path); });

    // replay in file order
    std::map<std::string, int> mat_map;
    ObjModel cur;
    cur.name = "unnamed_object";
    auto flush = [&]() {
        if (!cur.vi.empty()) out.models.push_back(cur);
        cur.vi.clear();
        cur.ni.clear();
    };
    for (ObjChunk& c : chunks) {
        size_t done = 0;
        auto take = [&](size_t upto) {
            cur.vi.insert(cur.vi.end(), c.vi.begin() + done, c.vi.begin() + upto);
            cur.ni.insert(cur.ni.end(), c.ni.begin() + done, c.ni.begin() + upto);
            done = upto;
        };
        for (const ObjEvent& ev : c.events) {
            take(ev.at);
            if (ev.kind == 'o') {
                flush();
                cur.name = ev.arg;
            } else if (ev.kind == 'u') {
                auto it = mat_map.find(ev.arg);
                const int m = it == mat_map.end() ? -1 : it->second;
                if (m != cur.material_id && !cur.vi.empty()) flush();
                cur.material_id = m;
            } else {
                load_mtl_names(dir + ev.arg, out, mat_map);
            }
        }
        take(c.vi.size());
        out.n_faces += c.n_faces;
        if (!c.error.empty()) return c.error;  // (a single pass stops at the first bad line as well)
        c.vi.clear(); c.vi.shrink_to_fit();
        c.ni.clear(); c.ni.shrink_to_fit();
    }
    flush();
    return std::string();
}

}  // namespace rthost

// Wavefront OBJ / MTL reader with the semantics Mesh::from_file relies on (shape.rs:352-363:
// tobj::load_obj with single_index + triangulate + ignore_points + ignore_lines):
//   * a new model starts at every `o` / `g`, and at a `usemtl` that changes the material while
//     faces are pending;
//   * polygons are fanned from their first vertex: a b c d -> (a,b,c), (a,c,d);
//   * a model's material id is the index of its `usemtl` name in the MTL file's newmtl order;
//   * `mtllib` is resolved relative to the OBJ file's directory.
// Only positions and normals are kept (texture coordinates are never read by the renderer).

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

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

std::string load_obj(const std::string& path, ObjData& out) {
    out = ObjData();
    std::string text;
    if (!read_file(path, text)) return "cannot read " + path;
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash + 1);

    std::map<std::string, int> mat_map;
    ObjModel cur;
    cur.name = "unnamed_object";
    auto flush = [&]() {
        if (!cur.vi.empty()) out.models.push_back(cur);
        cur.vi.clear();
        cur.ni.clear();
    };
    std::vector<long> fv, fn;
    const char* p = text.c_str();
    while (*p) {
        const char* q = skip_ws(p);
        if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
            char* e;
            for (int k = 0; k < 3; k++) { out.positions.push_back(std::strtof(q + 1, &e)); q = e - 1; }
        } else if (q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) {
            char* e;
            q += 2;
            for (int k = 0; k < 3; k++) { out.normals.push_back(std::strtof(q, &e)); q = e; }
        } else if (q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
            fv.clear(); fn.clear();
            q += 1;
            const long nv = (long)(out.positions.size() / 3), nn = (long)(out.normals.size() / 3);
            for (;;) {
                q = skip_ws(q);
                if (is_eol(*q)) break;
                char* e;
                long v = std::strtol(q, &e, 10);
                if (e == q) return "bad face in " + path;
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
                fv.push_back(v > 0 ? v - 1 : nv + v);
                fn.push_back(have_n ? (n > 0 ? n - 1 : nn + n) : -1);
            }
            if (fv.size() >= 3) {              // points and lines are ignored
                out.n_faces++;
                for (size_t i = 1; i + 1 < fv.size(); i++) {
                    const size_t c[3] = {0, i, i + 1};
                    for (size_t k : c) {
                        if (fv[k] < 0 || fv[k] >= nv) return "face index out of range in " + path;
                        if (fn[k] < 0 || fn[k] >= nn) return "face without a valid normal in " + path;
                        cur.vi.push_back((uint32_t)fv[k]);
                        cur.ni.push_back((uint32_t)fn[k]);
                    }
                }
            }
        } else if ((q[0] == 'o' || q[0] == 'g') && (q[1] == ' ' || q[1] == '\t' || is_eol(q[1]))) {
            flush();
            cur.name = rest_of_line(q + 1);
        } else if (std::strncmp(q, "usemtl", 6) == 0 && (q[6] == ' ' || q[6] == '\t')) {
            auto it = mat_map.find(first_token(q + 6));
            const int m = it == mat_map.end() ? -1 : it->second;
            if (m != cur.material_id && !cur.vi.empty()) flush();
            cur.material_id = m;
        } else if (std::strncmp(q, "mtllib", 6) == 0 && (q[6] == ' ' || q[6] == '\t')) {
            load_mtl_names(dir + first_token(q + 6), out, mat_map);
        }
        while (*p && *p != '\n') p++;
        if (*p) p++;
    }
    flush();
    return std::string();
}

}  // namespace rthost

// Catmull-Clark front end: quad / polygon control mesh (PLY) -> bicubic Bezier patches by feature-adaptive subdivision.
//
// Replaces, on the host (scene assembly):
//   CatmullClark::{load, new}                 reference src/primitive/catmull.rs:87-101
//   feature_adaptive_subdivision              catmull.rs:136-446   (face / edge / vertex points with semi-sharp creases,
//                                                                   edge split, face split, regular faces leave as patches)
//   check_if_regular                          catmull.rs:448-470
//   get_bezier_patch                          catmull.rs:472-549   (B-spline control net of a regular face -> Bezier)
//   get_gregory_patch + helpers               catmull.rs:551-764   (irregular faces left after the last iteration)
//
// The reference walks a half-edge mesh of the author's `pep-mesh` crate (a git dependency that is not vendored in
// /root/reference).  What is taken from it are the operations catmull.rs uses - their meaning follows from how the
// subdivision code wires new edges (catmull.rs:330-352): `vertex()` of a half-edge is its ORIGIN, `halfedge()` of a
// vertex leaves it, an edge's data (sharpness, the new edge point) is shared by its two half-edges, an open boundary is
// closed by faces flagged `is_boundary`.  What CANNOT be taken from it: which half-edge a freshly loaded face / vertex
// calls its first, and Rust's HashSet iteration order (catmull.rs:158-181).  Both only decide (a) the order of the
// patches, (b) which corner of a patch is (u, v) = (0, 0) and (c) the order in which four positions are summed.  The
// surface is the same; patch order, texcoord orientation and last-bit rounding are PARITY-UNPINNED against the
// reference.  Here faces and vertices start at the first half-edge the PLY lists and sets keep insertion order.
//
// The result is a list of patches; the loader turns every patch into an instance of its own (SPT_PRIM_BEZIER), so the
// scene's TLAS is the reference's BvhAccel<CubicBezier> (catmull.rs:445) and no new primitive kind crosses the ABI.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "hmath.hpp"
#include "host_scene.hpp"

namespace spt_host {

namespace {

constexpr int kNone = -1;

struct HalfEdgeMesh {
    struct Vert { V3 pos; bool has_new = false; V3 new_pos; int he = kNone; };
    struct Edge { bool has_new_pos = false; V3 new_pos; int new_vert = kNone; float sharpness = 0.0f; };
    struct Face { int he = kNone; bool boundary = false; bool has_new = false; V3 new_pos; bool is_regular = false; };
    struct Half { int next = kNone, vert = kNone, face = kNone; };   // half-edge h: twin = h ^ 1, edge = h >> 1
    std::vector<Vert> v;
    std::vector<Edge> e;
    std::vector<Face> f;
    std::vector<Half> h;

    int twin(int he) const { return he ^ 1; }
    Edge& edge(int he) { return e[(size_t)he >> 1]; }
    const Edge& edge(int he) const { return e[(size_t)he >> 1]; }
    int last(int he) const {
        int p = he;
        while (h[p].next != he) p = h[p].next;
        return p;
    }
    bool on_boundary(int he) const { return f[h[he].face].boundary; }
    int create_vertex(V3 pos) {
        Vert nv;
        nv.pos = pos;
        v.push_back(nv);
        return (int)v.size() - 1;
    }
    // halfedge::create_edge(a, b): (.0 from a to b, .1 from b to a)
    int create_edge(int a, int b, float sharpness) {
        Edge ne;
        ne.sharpness = sharpness;
        e.push_back(ne);
        Half h0, h1;
        h0.vert = a;
        h1.vert = b;
        h.push_back(h0);
        h.push_back(h1);
        return (int)h.size() - 2;
    }
    int create_face(bool boundary) {
        Face nf;
        nf.boundary = boundary;
        f.push_back(nf);
        return (int)f.size() - 1;
    }
    int vert_degree(int vi) const {
        int n = 0, he = v[vi].he;
        do { ++n; he = h[twin(he)].next; } while (he != v[vi].he);
        return n;
    }
    bool vert_on_boundary(int vi) const {
        int he = v[vi].he;
        do { if (on_boundary(he)) return true; he = h[twin(he)].next; } while (he != v[vi].he);
        return false;
    }
    int face_degree(int fi) const {
        int n = 0, he = f[fi].he;
        do { ++n; he = h[he].next; } while (he != f[fi].he);
        return n;
    }
};

[[noreturn]] void bad(const std::string& path, const std::string& what) { throw HostError(SPT_HOST_ERR_PARSE, "ply '" + path + "': " + what); }

// ply::load_to_halfedge: ASCII PLY with `vertex` (x y z, any further properties are skipped), `face` (one index list)
// and optionally `edge` (vertex1 vertex2 sharpness) elements.
HalfEdgeMesh load_ply(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw HostError(SPT_HOST_ERR_IO, "cannot open '" + path + "'");
    std::string line;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") bad(path, "not a PLY file");
    struct Elem { std::string name; size_t count = 0; std::vector<std::string> props; };
    std::vector<Elem> elems;
    bool ascii = false, header_done = false;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tok;
        ls >> tok;
        if (tok == "format") { std::string fmt; ls >> fmt; ascii = fmt == "ascii"; }
        else if (tok == "element") {
            Elem el;
            double cnt = -1.0;
            if (!(ls >> el.name >> cnt) || !(cnt >= 0.0) || cnt > 1e9 || cnt != std::floor(cnt)) bad(path, "bad element line '" + line + "'");
            el.count = (size_t)cnt;
            elems.push_back(el);
        }
        else if (tok == "property" && !elems.empty()) { std::string rest; std::getline(ls, rest); elems.back().props.push_back(rest); }
        else if (tok == "end_header") { header_done = true; break; }
    }
    if (!header_done) bad(path, "no end_header");
    if (!ascii) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "ply '" + path + "': only the ascii format is read");
    std::vector<V3> pos;
    std::vector<std::vector<int>> faces;
    struct Crease { int a, b; float s; };
    std::vector<Crease> creases;
    for (const Elem& el : elems) {
        // column of a named scalar property
        auto col = [&](const char* name) {
            for (size_t k = 0; k < el.props.size(); ++k) {
                std::istringstream ps(el.props[k]);
                std::string ty, nm;
                ps >> ty;
                if (ty == "list") return -1;
                ps >> nm;
                if (nm == name) return (int)k;
            }
            return -1;
        };
        for (size_t i = 0; i < el.count; ++i) {
            if (!std::getline(in, line)) bad(path, "unexpected end of file in element '" + el.name + "'");
            std::istringstream ls(line);
            std::vector<double> vals;
            double x;
            while (ls >> x) vals.push_back(x);
            if (el.name == "vertex") {
                const int cx = col("x"), cy = col("y"), cz = col("z");
                auto at = [&](int c) { return (c >= 0 && (size_t)c < vals.size()) ? (float)vals[(size_t)c] : 0.0f; };
                pos.push_back(V3{at(cx), at(cy), at(cz)});
            } else if (el.name == "face") {
                // the count is checked as a double BEFORE any cast: a value such as 1e30 converts to 0 on x86-64 and
                // would pass every integer comparison (the empty face then indexed an empty vector)
                const double cnt = vals.empty() ? -1.0 : vals[0];
                if (!(cnt >= 3.0) || cnt != std::floor(cnt) || cnt > (double)(vals.size() - 1)) bad(path, "bad face record");
                const size_t nfv = (size_t)cnt;
                std::vector<int> fv;
                for (size_t k = 0; k < nfv; ++k) {
                    const double dv = vals[1 + k];
                    if (!(dv >= 0.0) || dv != std::floor(dv) || dv >= (double)pos.size()) bad(path, "face index out of range");
                    fv.push_back((int)dv);
                }
                if (fv.size() < 3) bad(path, "bad face record");
                faces.push_back(fv);
            } else if (el.name == "edge") {
                const int c1 = col("vertex1"), c2 = col("vertex2"), cs = col("sharpness");
                if (c1 < 0 || c2 < 0 || (size_t)std::max(c1, c2) >= vals.size()) bad(path, "bad edge record");
                for (int c : {c1, c2}) {
                    const double dv = vals[(size_t)c];
                    if (!(dv >= 0.0) || dv != std::floor(dv) || dv > 2147483647.0) bad(path, "edge vertex out of range");
                }
                creases.push_back(Crease{(int)vals[(size_t)c1], (int)vals[(size_t)c2], (cs >= 0 && (size_t)cs < vals.size()) ? (float)vals[(size_t)cs] : 0.0f});
            }
        }
    }
    if (pos.empty() || faces.empty()) bad(path, "no vertices or no faces");
    HalfEdgeMesh m;
    for (const V3& p : pos) m.create_vertex(p);
    std::map<std::pair<int, int>, int> half_of;   // (origin, destination) -> half-edge
    for (const auto& fv : faces) {
        const int fi = m.create_face(false);
        std::vector<int> hs;
        for (size_t k = 0; k < fv.size(); ++k) {
            const int a = fv[k], b = fv[(k + 1) % fv.size()];
            if (a == b) bad(path, "degenerate face edge");
            if (half_of.count({a, b})) bad(path, "non-manifold mesh: an edge is used twice in the same direction");
            int he;
            auto tw = half_of.find({b, a});
            if (tw != half_of.end()) he = m.twin(tw->second);
            else he = m.create_edge(a, b, 0.0f);
            half_of[{a, b}] = he;
            m.h[he].vert = a;
            m.h[he].face = fi;
            if (m.v[a].he == kNone) m.v[a].he = he;
            hs.push_back(he);
        }
        for (size_t k = 0; k < hs.size(); ++k) m.h[hs[k]].next = hs[(k + 1) % hs.size()];
        m.f[fi].he = hs[0];
    }
    // open boundaries: loops of half-edges without a face are closed by boundary faces
    for (int he = 0; he < (int)m.h.size(); ++he) {
        if (m.h[he].face != kNone) continue;
        const int bf = m.create_face(true);
        int cur = he, guard = 0;
        do {
            m.h[cur].face = bf;
            // next boundary half-edge leaves the vertex this one arrives at: rotate around it
            const int dest = m.h[m.twin(cur)].vert;
            int nx = m.twin(cur);
            while ((m.h[nx].face != kNone && nx != he) || m.h[nx].vert != dest) {   // (the loop's first half-edge already carries bf)
                nx = m.twin(m.last(nx));   // previous half-edge of nx's face arrives at dest; its twin leaves dest
                if (++guard > 1 << 22) bad(path, "cannot close the mesh boundary (non-manifold vertex)");
            }
            m.h[cur].next = nx;
            cur = nx;
        } while (cur != he);
        m.f[bf].he = he;
    }
    for (size_t vi = 0; vi < m.v.size(); ++vi)
        if (m.v[vi].he == kNone) bad(path, "a vertex is not used by any face");
    for (const Crease& c : creases) {
        auto it = half_of.find({c.a, c.b});
        if (it == half_of.end()) it = half_of.find({c.b, c.a});
        if (it == half_of.end()) bad(path, "an edge record names two vertices that no face connects");
        m.edge(it->second).sharpness = c.s;
    }
    return m;
}

// catmull.rs:448-470
bool check_if_regular(const HalfEdgeMesh& m, int face) {
    if (m.face_degree(face) != 4) return false;
    int he = m.f[face].he;
    do {
        const int vi = m.h[he].vert;
        int deg = m.vert_degree(vi);
        if (m.vert_on_boundary(vi)) deg += 1;
        if (deg != 4 || m.edge(he).sharpness > 0.0f) return false;
        he = m.h[he].next;
    } while (he != m.f[face].he);
    return true;
}

struct Patch { V3 cp[4][4]; };

// catmull.rs:472-549
Patch get_bezier_patch(const HalfEdgeMesh& m, int face) {
    Patch out;
    V3 cp[4][4];
    static const int order[4][4][2] = {
        {{1, 1}, {0, 1}, {1, 0}, {0, 0}},
        {{1, 2}, {1, 3}, {0, 2}, {0, 3}},
        {{2, 2}, {3, 2}, {2, 3}, {3, 3}},
        {{2, 1}, {2, 0}, {3, 1}, {3, 0}},
    };
    auto pos = [&](int he) { return m.v[m.h[he].vert].pos; };
    auto nx = [&](int he) { return m.h[he].next; };
    auto tw = [&](int he) { return m.twin(he); };
    int he = m.f[face].he;
    for (int c = 0; c < 4; ++c) {
        const int last = he;
        he = nx(he);
        V3& p0 = cp[order[c][0][0]][order[c][0][1]];
        V3& p1 = cp[order[c][1][0]][order[c][1][1]];
        V3& p2 = cp[order[c][2][0]][order[c][2][1]];
        V3& p3 = cp[order[c][3][0]][order[c][3][1]];
        p0 = pos(he);
        if (m.on_boundary(tw(he))) {
            p1 = p0 + (p0 - pos(last));
            const int he2 = tw(nx(tw(he)));
            p2 = pos(he2);
            p3 = p2 + (p2 - pos(m.last(he2)));
        } else if (m.on_boundary(tw(last))) {
            const int he2 = tw(nx(tw(he)));
            p1 = pos(he2);
            p2 = p0 + (p0 - pos(tw(he)));
            p3 = p1 + (p1 - pos(tw(nx(tw(he2)))));
        } else {
            int he2 = tw(nx(tw(he)));
            p1 = pos(he2);
            he2 = nx(nx(he2));
            p2 = pos(he2);
            he2 = nx(he2);
            p3 = pos(he2);
        }
    }
    static const float tm[4][4] = {
        {1.0f / 6.0f, 4.0f / 6.0f, 1.0f / 6.0f, 0.0f},
        {0.0f, 4.0f / 6.0f, 2.0f / 6.0f, 0.0f},
        {0.0f, 2.0f / 6.0f, 4.0f / 6.0f, 0.0f},
        {0.0f, 1.0f / 6.0f, 4.0f / 6.0f, 1.0f / 6.0f},
    };
    V3 tmp[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            tmp[i][j] = V3{0, 0, 0};
            for (int k = 0; k < 4; ++k) tmp[i][j] = tmp[i][j] + cp[i][k] * tm[j][k];
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            out.cp[j][i] = V3{0, 0, 0};
            for (int k = 0; k < 4; ++k) out.cp[j][i] = out.cp[j][i] + tmp[k][i] * tm[j][k];
        }
    return out;
}

// catmull.rs:620-662: edge mid points and face centroids around `vert`, starting behind `face`, reversed
void edge_and_face_points(const HalfEdgeMesh& m, int vert, int face, std::vector<V3>& ep, std::vector<V3>& fp) {
    ep.clear();
    fp.clear();
    int he = m.v[vert].he;
    while (m.h[he].face != face) he = m.h[m.twin(he)].next;
    he = m.h[m.twin(he)].next;
    const int start = he;
    const V3 pv = m.v[vert].pos;
    do {
        const int twin = m.twin(he);
        ep.push_back((m.v[m.h[twin].vert].pos + pv) * 0.5f);
        V3 pf{0, 0, 0};
        float cnt = 0.0f;
        int fhe = he;
        do { pf = pf + m.v[m.h[fhe].vert].pos; cnt += 1.0f; fhe = m.h[fhe].next; } while (fhe != he);
        fp.push_back(pf / cnt);
        he = m.h[twin].next;
    } while (he != start);
    std::vector<V3> re(ep.rbegin(), ep.rend()), rf(fp.rbegin(), fp.rend());
    ep.swap(re);
    fp.swap(rf);
}
constexpr float kPi = 3.14159265358979323846f;
// catmull.rs:664-685
V3 vertex_control_point(V3 pv, const std::vector<V3>& ep, const std::vector<V3>& fp) {
    V3 sum{0, 0, 0};
    for (const V3& p : ep) sum = sum + p;
    for (const V3& p : fp) sum = sum + p;
    const float n = (float)ep.size(), n_inv = 1.0f / n, n5_inv = 1.0f / (n + 5.0f);
    return pv * ((n - 3.0f) * n5_inv) + sum * (4.0f * n_inv * n5_inv);
}
// catmull.rs:687-726
void edge_control_points(V3 pv, const std::vector<V3>& ep, const std::vector<V3>& fp, V3* e_pos, V3* e_neg) {
    const float n = (float)ep.size(), n_inv = 1.0f / n;
    const float frac_pi_n = kPi * n_inv, c_pi_n = std::cos(frac_pi_n), frac_2pi_n = 2.0f * kPi * n_inv;
    const float sigma = 1.0f / std::sqrt(4.0f + c_pi_n * c_pi_n);
    const float temp = std::cos(frac_2pi_n);
    const float lambda = (5.0f + temp + c_pi_n * std::sqrt(18.0f + 2.0f * temp)) / 24.0f;
    V3 tangent{0, 0, 0}, bitangent{0, 0, 0};
    const float ka_common = 1.0f - sigma * c_pi_n, kb_common = 2.0f * sigma;
    for (size_t i = 0; i < ep.size(); ++i) {
        const float ti = (float)i;
        float ka = ka_common * std::cos(frac_2pi_n * ti), kb = kb_common * std::cos(frac_2pi_n * ti + frac_pi_n);
        tangent = tangent + (ep[i] * ka + fp[i] * kb);
        const float bi = ti - 1.0f;
        ka = ka_common * std::cos(frac_2pi_n * bi);
        kb = kb_common * std::cos(frac_2pi_n * bi + frac_pi_n);
        bitangent = bitangent + (ep[i] * ka + fp[i] * kb);
    }
    tangent = tangent * 2.0f * n_inv;
    bitangent = bitangent * 2.0f * n_inv;
    *e_pos = pv + tangent * lambda;
    *e_neg = pv + bitangent * lambda;
}
// catmull.rs:728-764
V3 face_control_point_pos(V3 pos0, V3 e0_pos, V3 e1_neg, const std::vector<V3>& ep, const std::vector<V3>& fp, float n0, float n1) {
    const V3 r = (ep[ep.size() - 1] - ep[1]) / 3.0f + (fp[0] - fp[fp.size() - 1]) * 2.0f / 3.0f;
    const float c0 = std::cos(2.0f * kPi / n0), c1 = std::cos(2.0f * kPi / n1);
    return (pos0 * c1 + e0_pos * (3.0f - 2.0f * c0 - c1) + e1_neg * (2.0f * c0) + r) / 3.0f;
}
V3 face_control_point_neg(V3 pos0, V3 e0_neg, V3 e3_pos, const std::vector<V3>& ep, const std::vector<V3>& fp, float n0, float n3) {
    const V3 r = (ep[0] - ep[2]) / 3.0f + (fp[0] - fp[1]) * 2.0f / 3.0f;
    const float c0 = std::cos(2.0f * kPi / n0), c1 = std::cos(2.0f * kPi / n3);
    return (pos0 * c1 + e0_neg * (3.0f - 2.0f * c0 - c1) + e3_pos * (2.0f * c0) + r) / 3.0f;
}

// catmull.rs:551-618: an irregular quad left after the last iteration, as the bicubic patch whose interior points are
// the averages of the Gregory patch's pairs
Patch get_gregory_patch(const HalfEdgeMesh& m, int face) {
    Patch out;
    std::vector<V3> ep[4], fp[4];
    V3 pos[4], e_pos[4], e_neg[4];
    int he = m.f[face].he;
    for (int k = 0; k < 4; ++k) {
        const int vert = m.h[he].vert;
        edge_and_face_points(m, vert, face, ep[k], fp[k]);
        if (ep[k].size() < 3) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "catmull_clark: a vertex of valence < 3 on an irregular face");
        pos[k] = vertex_control_point(m.v[vert].pos, ep[k], fp[k]);
        edge_control_points(pos[k], ep[k], fp[k], &e_pos[k], &e_neg[k]);
        he = m.h[he].next;
    }
    out.cp[0][0] = pos[0]; out.cp[0][1] = e_pos[0]; out.cp[1][0] = e_neg[0];
    out.cp[0][3] = pos[1]; out.cp[1][3] = e_pos[1]; out.cp[0][2] = e_neg[1];
    out.cp[3][3] = pos[2]; out.cp[3][2] = e_pos[2]; out.cp[2][3] = e_neg[2];
    out.cp[3][0] = pos[3]; out.cp[2][0] = e_pos[3]; out.cp[3][1] = e_neg[3];
    const float n[4] = {(float)ep[0].size(), (float)ep[1].size(), (float)ep[2].size(), (float)ep[3].size()};
    const int to[4][2] = {{1, 1}, {1, 2}, {2, 2}, {2, 1}};
    for (int k = 0; k < 4; ++k) {
        const int k1 = (k + 1) % 4, k3 = (k + 3) % 4;
        const V3 fpos = face_control_point_pos(pos[k], e_pos[k], e_neg[k1], ep[k], fp[k], n[k], n[k1]);
        const V3 fneg = face_control_point_neg(pos[k], e_neg[k], e_pos[k3], ep[k], fp[k], n[k], n[k3]);
        out.cp[to[k][0]][to[k][1]] = (fpos + fneg) * 0.5f;
    }
    return out;
}

// catmull.rs:136-446
std::vector<Patch> feature_adaptive_subdivision(HalfEdgeMesh& m, uint32_t max_iter_times) {
    std::vector<int> process_faces;
    for (int fi = 0; fi < (int)m.f.size(); ++fi) process_faces.push_back(fi);
    std::vector<Patch> patches;
    for (uint32_t iter = 0; iter < max_iter_times; ++iter) {
        std::vector<int> irregular;
        for (int face : process_faces) {
            if (m.f[face].boundary) continue;
            const bool reg = check_if_regular(m, face);
            m.f[face].is_regular = reg;
            if (!reg) irregular.push_back(face);
            else patches.push_back(get_bezier_patch(m, face));
        }
        // every face around a vertex of an irregular face is subdivided (the reference's HashSet: here in insertion order)
        std::vector<int> tbs;
        std::set<int> seen;
        for (int face : irregular) {
            int he = m.f[face].he;
            do {
                int vhe = he;
                do {
                    const int f = m.h[vhe].face;
                    if (!m.f[f].boundary && seen.insert(f).second) tbs.push_back(f);
                    vhe = m.h[m.twin(vhe)].next;
                } while (vhe != he);
                he = m.h[he].next;
            } while (he != m.f[face].he);
        }
        // face points
        for (int face : tbs) {
            float count = 0.0f;
            V3 sum{0, 0, 0};
            int he = m.f[face].he;
            do { count += 1.0f; sum = sum + m.v[m.h[he].vert].pos; he = m.h[he].next; } while (he != m.f[face].he);
            m.f[face].has_new = true;
            m.f[face].new_pos = sum / count;
        }
        // edge points
        for (int face : tbs) {
            int he = m.f[face].he;
            do {
                HalfEdgeMesh::Edge& ed = m.edge(he);
                if (!ed.has_new_pos) {
                    const int twin = m.twin(he);
                    const V3 p1 = m.v[m.h[he].vert].pos, p2 = m.v[m.h[twin].vert].pos;
                    const float sharp = ed.sharpness;
                    const V3 crease = (p1 + p2) * 0.5f;
                    const HalfEdgeMesh::Face &f1 = m.f[m.h[he].face], &f2 = m.f[m.h[twin].face];
                    const V3 smooth = (f1.has_new && f2.has_new) ? (p1 + p2 + f1.new_pos + f2.new_pos) * 0.25f : (p1 + p2) * 0.5f;
                    if (m.on_boundary(he) || m.on_boundary(twin) || sharp >= 1.0f) ed.new_pos = crease;
                    else if (sharp > 0.0f) ed.new_pos = crease * sharp + smooth * (1.0f - sharp);
                    else ed.new_pos = smooth;
                    ed.has_new_pos = true;
                }
                he = m.h[he].next;
            } while (he != m.f[face].he);
        }
        // vertex points
        for (int face : tbs) {
            int he = m.f[face].he;
            do {
                const int vi = m.h[he].vert;
                if (!m.v[vi].has_new) {
                    int num_creases = 0;
                    V3 c1{0, 0, 0}, c2{0, 0, 0};
                    int vhe = m.v[vi].he;
                    do {
                        const int twin = m.twin(vhe);
                        if (m.edge(vhe).sharpness > 0.0f || m.on_boundary(vhe) || m.on_boundary(twin)) {
                            ++num_creases;
                            if (num_creases == 1) c1 = m.v[m.h[twin].vert].pos;
                            else if (num_creases == 2) c2 = m.v[m.h[twin].vert].pos;
                        }
                        vhe = m.h[twin].next;
                    } while (vhe != m.v[vi].he);
                    const V3 pv = m.v[vi].pos;
                    V3 np;
                    if (num_creases > 2) {
                        np = pv;
                    } else if (num_creases == 2) {
                        np = pv * 0.75f + c1 * 0.125f + c2 * 0.125f;
                    } else {
                        float n = 0.0f;
                        V3 sum{0, 0, 0};
                        int w = he;
                        do {
                            const int twin = m.twin(w);
                            n += 1.0f;
                            sum = sum + m.v[m.h[twin].vert].pos;
                            const HalfEdgeMesh::Face& wf = m.f[m.h[w].face];
                            sum = sum + (wf.has_new ? wf.new_pos : pv);
                            w = m.h[twin].next;
                        } while (w != he);
                        const float n_inv = 1.0f / n;
                        np = (pv * (n - 2.0f) + sum * n_inv) * n_inv;
                    }
                    m.v[vi].has_new = true;
                    m.v[vi].new_pos = np;
                }
                he = m.h[he].next;
            } while (he != m.f[face].he);
        }
        // split edges
        std::vector<int> split;
        for (int face : tbs) {
            int he = m.f[face].he;
            do {
                if (m.edge(he).new_vert == kNone) {
                    m.edge(he).new_vert = m.create_vertex(m.edge(he).new_pos);
                    split.push_back(he);
                }
                he = m.h[he].next;
            } while (he != m.f[face].he);
        }
        for (int he : split) {
            const int ev = m.edge(he).new_vert;
            const float ns = std::max(m.edge(he).sharpness - 1.0f, 0.0f);
            m.edge(he) = HalfEdgeMesh::Edge();
            m.edge(he).sharpness = ns;
            const int ne0 = m.create_edge(m.h[he].vert, ev, ns), ne1 = ne0 + 1;
            const int twin = m.twin(he);
            m.h[ne0].next = he;
            m.h[m.last(he)].next = ne0;
            m.v[m.h[he].vert].he = ne0;
            m.h[he].vert = ev;
            m.h[ne0].face = m.h[he].face;
            m.f[m.h[he].face].he = ne0;
            m.h[ne1].next = m.h[twin].next;
            m.h[twin].next = ne1;
            m.h[ne1].face = m.h[twin].face;
            m.f[m.h[twin].face].he = twin;
            m.v[ev].he = he;
        }
        // split faces
        process_faces.clear();
        for (int face : tbs) {
            const int fv = m.create_vertex(m.f[face].new_pos);
            const bool was_regular = m.f[face].is_regular, was_boundary = m.f[face].boundary;
            const int first_he = m.f[face].he;
            m.f[face] = HalfEdgeMesh::Face();
            m.f[face].he = first_he;
            m.f[face].boundary = was_boundary;
            std::vector<int> new_edges, new_faces;
            int count = 0;
            int he = first_he;
            do {
                const int he_last = he;
                he = m.h[he].next;
                const int ev = m.h[he].vert, he_next = he;
                const int ne0 = m.create_edge(ev, fv, 0.0f), ne1 = ne0 + 1;
                m.v[fv].he = ne1;
                m.h[ne1].next = he_next;
                m.h[he_last].next = ne0;
                new_edges.push_back(ne0);
                new_edges.push_back(ne1);
                new_faces.push_back(count == 0 ? face : m.create_face(was_boundary));
                ++count;
                HalfEdgeMesh::Vert& vv = m.v[m.h[he_last].vert];
                if (vv.has_new) { vv.pos = vv.new_pos; vv.has_new = false; }
                he = m.h[he].next;
            } while (he != first_he);
            for (int i = 0; i < count; ++i) {
                const int j = i == 0 ? 2 * count - 1 : 2 * i - 1;
                const int ej = new_edges[(size_t)j];
                m.h[new_edges[(size_t)(2 * i)]].next = ej;
                m.f[new_faces[(size_t)i]].he = ej;
                int w = new_edges[(size_t)(2 * i)];
                do { m.h[w].face = new_faces[(size_t)i]; w = m.h[w].next; } while (w != new_edges[(size_t)(2 * i)]);
            }
            if (!was_regular) process_faces.insert(process_faces.end(), new_faces.begin(), new_faces.end());
        }
    }
    for (int face : process_faces) {
        if (m.f[face].boundary) continue;
        if (check_if_regular(m, face)) patches.push_back(get_bezier_patch(m, face));
        else if (m.face_degree(face) == 4) patches.push_back(get_gregory_patch(m, face));
        else throw HostError(SPT_HOST_ERR_UNSUPPORTED, "catmull_clark: fas_times = 0 with a face that is not a quad (the reference reads four corners of any face)");
    }
    return patches;
}

}  // namespace

// CatmullClark::load (catmull.rs:93-101): 16 control points (x, y, z) per patch
std::vector<float> catmull_clark_patches(const std::string& ply_path, uint32_t fas_times) {
    HalfEdgeMesh mesh = load_ply(ply_path);
    std::vector<Patch> patches = feature_adaptive_subdivision(mesh, fas_times);
    std::vector<float> out;
    out.reserve(patches.size() * 48);
    for (const Patch& p : patches)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) { out.push_back(p.cp[i][j].x); out.push_back(p.cp[i][j].y); out.push_back(p.cp[i][j].z); }
    return out;
}

}  // namespace spt_host

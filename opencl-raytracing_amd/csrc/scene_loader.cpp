// scene_loader.cpp -- scene files -> mrt_scene (SoA, FP32).
//
// Same schema, validation order and messages as the reference loader
// rayson::read_json / read_file (rayson.hpp:380-638), followed by the
// conversion gfx::scene::read_json performs (gfxraytrace.hpp:645-721):
// colours to float, vectors kept in double, camera basis built in double
// (gfxraytrace.hpp:881-889) and only then cast to float (:727-808); objects
// ordered spheres first, then triangles (:701-714).  Constructor arguments in
// the reference are checked left to right (the order its clang build uses).
#include "mrt.h"
#include "json_min.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

using jsonmin::Value;

namespace {

// A scene file that the schema rejects; the message is what the caller sees.
struct Rejected { std::string message; };

struct Vec3 { double x = 0, y = 0, z = 0; };

// What a float-typed field has to satisfy beyond being a float token.
enum class Range { Any, Positive, Negative, NonNegative };

// One JSON object seen through the scene schema.  Every accessor checks
// presence first, then the token type, then the range, and rejects with the
// message the reference gives for that failure (rayson.hpp:390-487).
class Fields {
public:
    explicit Fields(const Value& object) : obj_(object) {}

    bool present(const char* key) const { return obj_.find(key) != nullptr; }
    const Value* child(const char* key) const { return obj_.find(key); }

    bool flag(const char* key) const {
        const Value& v = at(key);
        if (!v.is_boolean()) reject(quoted(key) + " must be a bool");               // :396-402
        return v.b;
    }
    // Only float-typed tokens pass: 20 is rejected where 20.0 is accepted (:404-410, :451-473).
    double real(const char* key, Range range = Range::Any) const {
        const Value& v = at(key);
        if (!v.is_number_float()) reject(quoted(key) + " must be a float");
        const double x = v.d;
        switch (range) {
            case Range::Positive:    if (x <= 0.0) reject(quoted(key) + " must be positive"); break;
            case Range::Negative:    if (x >= 0.0) reject(quoted(key) + " must be negative"); break;
            case Range::NonNegative: if (x < 0.0) reject(quoted(key) + " must be non-negative"); break;
            case Range::Any: break;
        }
        return x;
    }
    std::string text(const char* key) const {
        const Value& v = at(key);
        if (!v.is_string()) reject(quoted(key) + " must be a string");               // :412-418
        return v.s;
    }
    unsigned count(const char* key) const {                                          // :420-430
        const Value& v = at(key);
        if (!v.is_number_integer()) reject(quoted(key) + " must be an integer");
        const int x = (v.kind == jsonmin::Kind::Int) ? static_cast<int>(v.i) : static_cast<int>(v.u);
        if (x <= 0) reject(quoted(key) + " must be positive");
        return static_cast<unsigned>(x);
    }
    Vec3 vec3(const char* key) const {                                               // :432-449
        const Value& a = at(key);
        if (!a.is_array()) reject(std::string("expected ") + key + " to be an array");
        // the count reported is the size of the enclosing object, as at :438-441
        if (a.size() != 3)
            reject(std::string("expected array ") + key + " to have 3 elements, but found " +
                   std::to_string(obj_.size()) + " elements");
        double c[3];
        bool all = true;
        for (int k = 0; k < 3; ++k) {
            bool ok = false;
            c[k] = a.items[k].as_double(ok);
            all = all && ok;
        }
        if (!all) reject("vector3 must contain numbers");
        return Vec3{c[0], c[1], c[2]};
    }
    Vec3 color(const char* key) const {                                              // :475-487
        const Vec3 c = vec3(key);
        const double comp[3] = {c.x, c.y, c.z};
        for (int k = 0; k < 3; ++k)
            if (comp[k] < 0.0 || comp[k] > 1.0)
                reject(std::string("color has ") + "rgb"[k] + " component outside the range [0, 1]");
        return c;
    }

    [[noreturn]] static void reject(const std::string& message) { throw Rejected{message}; }

private:
    static std::string quoted(const char* key) { return std::string("key \"") + key + "\""; }
    const Value& at(const char* key) const {
        const Value* v = obj_.find(key);
        if (!v) reject(std::string("missing ") + quoted(key));                       // :390-394
        return *v;
    }
    const Value& obj_;
};

}  // namespace

struct mrt_loaded_scene {
    mrt_scene view{};
    mrt_scene_info info{};
    std::vector<float> sph[4];
    std::vector<uint32_t> sph_mat;
    std::vector<float> tri[9];
    std::vector<uint32_t> tri_mat;
    std::vector<float> mat[4];
    std::vector<float> light[6];
    std::vector<double> light_pos, light_intensity;
    std::vector<std::string> material_names;
};

namespace {

template <size_t N>
void append(std::vector<float> (&planes)[N], size_t first, const Vec3& v) {
    planes[first + 0].push_back(static_cast<float>(v.x));
    planes[first + 1].push_back(static_cast<float>(v.y));
    planes[first + 2].push_back(static_cast<float>(v.z));
}

void store3(float (&dst)[3], const double (&src)[3]) {
    for (int k = 0; k < 3; ++k) dst[k] = static_cast<float>(src[k]);
}

// The element objects of an optional or required list, in file order.
std::vector<const Value*> elements(const Fields& scene, const char* key) {
    const Value* list = scene.child(key);
    return list ? list->children() : std::vector<const Value*>();
}

// One pass over the document in the reference's order of checks (rayson.hpp:380-618); values
// go straight into the FP32 planes of `out` (the double -> float casts of
// gfxraytrace.hpp:727-808), objects as spheres first, then triangles (:701-714).
void build(const Value& document, mrt_loaded_scene& out) {
    if (!document.is_object()) Fields::reject("rayson must be comprised of one JSON object");   // :382-384
    const Fields scene(document);
    mrt_scene& s = out.view;
    mrt_scene_info& info = out.info;

    // camera (:489-491) and viewport (:493-498)
    const Vec3 eye = scene.vec3("camera_eye");
    const Vec3 up = scene.vec3("camera_up");
    const Vec3 view = scene.vec3("camera_view");
    s.xres = scene.count("x_resolution");
    s.yres = scene.count("y_resolution");
    info.left = scene.real("viewport_left", Range::Negative);
    info.top = scene.real("viewport_top", Range::Positive);
    info.right = scene.real("viewport_right", Range::Positive);
    info.bottom = scene.real("viewport_bottom", Range::Negative);

    // projection: exactly one of the two keys (:500-521)
    {
        const bool ortho = scene.present("ortho_projection"), persp = scene.present("persp_focal_length");
        if (ortho && persp) Fields::reject("cannot have both ortho_projection and persp_focal_length");
        if (!ortho && !persp) Fields::reject("must have ortho_projection or persp_focal_length");
        if (ortho) {
            if (!scene.flag("ortho_projection")) Fields::reject("ortho_projection, if present, must be true");
            s.projection = MRT_PROJ_ORTHO;
            info.focal = 0.0;
        } else {
            s.projection = MRT_PROJ_PERSP;
            info.focal = scene.real("persp_focal_length", Range::Positive);
        }
    }

    // shader: flat_shader wins when both are present (:523-540)
    Vec3 ambient;
    if (scene.present("flat_shader")) {
        if (!scene.flag("flat_shader")) Fields::reject("flat_shader, if present, must be true");
        s.shader = MRT_SHADER_FLAT;
    } else if (const Value* phong_obj = scene.child("phong_shader")) {
        const Fields phong(*phong_obj);
        s.shader = MRT_SHADER_PHONG;
        info.ka = phong.real("ambient_coeff", Range::NonNegative);
        info.kd = phong.real("diffuse_coeff", Range::NonNegative);
        info.ks = phong.real("specular_coeff", Range::NonNegative);
        ambient = phong.color("ambient_color");
    } else {
        Fields::reject("must have flat_shader or phong_shader");
    }
    const Vec3 background = scene.color("background");                               // :542

    // point lights, optional (:550-560)
    if (const Value* list = scene.child("point_lights")) {
        if (!list->is_array()) Fields::reject("expected point_lights to be an array");
        for (const Value* item : list->children()) {
            const Fields light(*item);
            const Vec3 at = light.vec3("location");
            append(out.light, 0, at);
            append(out.light, 3, light.color("color"));
            out.light_intensity.push_back(light.real("intensity", Range::Positive));
            for (double c : {at.x, at.y, at.z}) out.light_pos.push_back(c);
        }
    }

    // materials, required; names must be unique (:562-581)
    if (!scene.present("materials")) Fields::reject("no materials list");
    for (const Value* item : elements(scene, "materials")) {
        const Fields material(*item);
        out.material_names.push_back(material.text("name"));
        out.mat[3].push_back(static_cast<float>(material.real("shininess", Range::Positive)));
        append(out.mat, 0, material.color("color"));
    }
    std::unordered_map<std::string, uint32_t> index_of;
    for (uint32_t k = 0; k < out.material_names.size(); ++k)
        if (!index_of.emplace(out.material_names[k], k).second)
            Fields::reject("duplicate material name \"" + out.material_names[k] + "\"");
    const auto material_index = [&](const std::string& name, const char* who) {
        const auto found = index_of.find(name);
        if (found == index_of.end()) Fields::reject(std::string(who) + " references undefined material \"" + name + "\"");
        return found->second;
    };

    // spheres, optional: the material is resolved before the geometry is read (:583-594)
    for (const Value* item : elements(scene, "spheres")) {
        const Fields sphere(*item);
        out.sph_mat.push_back(material_index(sphere.text("material"), "sphere"));
        append(out.sph, 0, sphere.vec3("center"));
        out.sph[3].push_back(static_cast<float>(sphere.real("radius", Range::Positive)));
    }

    // triangles, optional: vertices are read and compared (in double) before the material is
    // resolved (:596-615)
    for (const Value* item : elements(scene, "triangles")) {
        const Fields triangle(*item);
        const std::string name = triangle.text("material");
        const Vec3 v[3] = {triangle.vec3("a"), triangle.vec3("b"), triangle.vec3("c")};
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 3; ++q)
                if (v[p].x == v[q].x && v[p].y == v[q].y && v[p].z == v[q].z)
                    Fields::reject("triangle is degenerate due to duplicated vertices");
        out.tri_mat.push_back(material_index(name, "triangle"));
        for (int p = 0; p < 3; ++p) append(out.tri, 3 * p, v[p]);
    }

    // Camera basis in double, cast to float only at the end (gfxraytrace.hpp:881-889, :727-734):
    // w = -view / |view|, u = (up x w) / |up x w|, v = w x u.
    const double vlen = std::sqrt(view.x * view.x + view.y * view.y + view.z * view.z);
    const double w[3] = {-(view.x / vlen), -(view.y / vlen), -(view.z / vlen)};
    const double uw[3] = {up.y * w[2] - up.z * w[1], up.z * w[0] - up.x * w[2], up.x * w[1] - up.y * w[0]};
    const double ulen = std::sqrt(uw[0] * uw[0] + uw[1] * uw[1] + uw[2] * uw[2]);
    const double u[3] = {uw[0] / ulen, uw[1] / ulen, uw[2] / ulen};
    const double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
    const double e[3] = {eye.x, eye.y, eye.z};
    for (int k = 0; k < 3; ++k) {
        info.eye[k] = e[k]; info.u[k] = u[k]; info.v[k] = v[k]; info.w[k] = w[k];
    }
    store3(s.eye, e); store3(s.u, u); store3(s.v, v); store3(s.w, w);
    s.left = static_cast<float>(info.left);
    s.right = static_cast<float>(info.right);
    s.top = static_cast<float>(info.top);
    s.bottom = static_cast<float>(info.bottom);
    s.focal = static_cast<float>(info.focal);
    s.ka = static_cast<float>(info.ka);
    s.kd = static_cast<float>(info.kd);
    s.ks = static_cast<float>(info.ks);
    const double amb[3] = {ambient.x, ambient.y, ambient.z}, bg[3] = {background.x, background.y, background.z};
    store3(s.ambient, amb);
    store3(s.background, bg);

    // the view the C ABI hands out points into the planes above
    s.n_spheres = static_cast<uint32_t>(out.sph_mat.size());
    s.n_tris = static_cast<uint32_t>(out.tri_mat.size());
    s.n_materials = static_cast<uint32_t>(out.material_names.size());
    s.n_lights = static_cast<uint32_t>(out.light_intensity.size());
    for (int k = 0; k < 4; ++k) s.sph[k] = out.sph[k].data();
    for (int k = 0; k < 9; ++k) s.tri[k] = out.tri[k].data();
    for (int k = 0; k < 4; ++k) s.mat[k] = out.mat[k].data();
    for (int k = 0; k < 6; ++k) s.light[k] = out.light[k].data();
    s.sph_mat = out.sph_mat.data();
    s.tri_mat = out.tri_mat.data();
    info.n_lights = s.n_lights;
    info.light_pos = out.light_pos.data();
    info.light_intensity = out.light_intensity.data();
}

void put_err(char* err, size_t cap, const std::string& msg) {
    if (!err || cap == 0) return;
    size_t n = msg.size() < cap - 1 ? msg.size() : cap - 1;
    std::memcpy(err, msg.data(), n);
    err[n] = 0;
}

int load_text(const char* text, size_t len, const std::string& path_for_msg, mrt_loaded_scene** out,
              char* err, size_t err_cap) {
    if (!out) return MRT_ERR_INVALID;
    *out = nullptr;
    Value root;
    std::string perr;
    if (!jsonmin::parse(text, len, root, perr)) {
        put_err(err, err_cap, "JSON parse error reading \"" + path_for_msg + "\"");   // rayson.hpp:630-633
        return MRT_ERR_PARSE;
    }
    auto* ls = new mrt_loaded_scene();
    try {
        build(root, *ls);
    } catch (const Rejected& e) {
        delete ls;
        put_err(err, err_cap, e.message);
        return MRT_ERR_PARSE;
    } catch (const std::exception& e) {
        delete ls;
        put_err(err, err_cap, std::string("internal error: ") + e.what());
        return MRT_ERR_INVALID;
    }
    *out = ls;
    put_err(err, err_cap, "");
    return MRT_OK;
}

}  // namespace

extern "C" {

int mrt_scene_load_file(const char* path, mrt_loaded_scene** out, char* err, size_t err_cap) {
    if (!path || !out) return MRT_ERR_INVALID;
    *out = nullptr;
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        put_err(err, err_cap, std::string("could not open \"") + path + "\"");        // rayson.hpp:622-625
        return MRT_ERR_IO;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    return load_text(text.data(), text.size(), path, out, err, err_cap);
}

int mrt_scene_load_text(const char* json_text, size_t len, mrt_loaded_scene** out, char* err,
                        size_t err_cap) {
    if (!json_text) return MRT_ERR_INVALID;
    return load_text(json_text, len, "<memory>", out, err, err_cap);
}

const mrt_scene* mrt_scene_view(const mrt_loaded_scene* s) { return s ? &s->view : nullptr; }
const mrt_scene_info* mrt_scene_info_view(const mrt_loaded_scene* s) { return s ? &s->info : nullptr; }
void mrt_scene_free(mrt_loaded_scene* s) { delete s; }

}  // extern "C"

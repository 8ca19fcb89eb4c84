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

struct ReadError { std::string message; };

struct Vec3 { double x = 0, y = 0, z = 0; };
inline bool same(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

bool has(const Value& obj, const std::string& key) { return obj.find(key) != nullptr; }

const Value& need(const Value& obj, const std::string& key) {
    const Value* v = obj.find(key);
    if (!v) throw ReadError{"missing key \"" + key + "\""};      // rayson.hpp:390-394
    return *v;
}

bool get_bool(const Value& obj, const std::string& key) {
    const Value& v = need(obj, key);
    if (!v.is_boolean()) throw ReadError{"key \"" + key + "\" must be a bool"};   // :396-402
    return v.b;
}

double get_double(const Value& obj, const std::string& key) {
    const Value& v = need(obj, key);
    // Only float-typed tokens pass: 20 is rejected where 20.0 is accepted (:404-410).
    if (!v.is_number_float()) throw ReadError{"key \"" + key + "\" must be a float"};
    return v.d;
}

std::string get_string(const Value& obj, const std::string& key) {
    const Value& v = need(obj, key);
    if (!v.is_string()) throw ReadError{"key \"" + key + "\" must be a string"};  // :412-418
    return v.s;
}

unsigned get_positive_unsigned(const Value& obj, const std::string& key) {
    const Value& v = need(obj, key);
    if (!v.is_number_integer()) throw ReadError{"key \"" + key + "\" must be an integer"};  // :420-430
    int x = (v.kind == jsonmin::Kind::Int) ? static_cast<int>(v.i) : static_cast<int>(v.u);
    if (x <= 0) throw ReadError{"key \"" + key + "\" must be positive"};
    return static_cast<unsigned>(x);
}

Vec3 get_vector3(const Value& obj, const std::string& key) {
    const Value& a = need(obj, key);
    if (!a.is_array()) throw ReadError{"expected " + key + " to be an array"};     // :435-437
    if (a.size() != 3) {
        // The count reported is the size of the enclosing object, as at :438-441.
        throw ReadError{"expected array " + key + " to have 3 elements, but found " +
                        std::to_string(obj.size()) + " elements"};
    }
    bool ok0, ok1, ok2;
    Vec3 r;
    r.x = a.items[0].as_double(ok0);
    r.y = a.items[1].as_double(ok1);
    r.z = a.items[2].as_double(ok2);
    if (!(ok0 && ok1 && ok2)) throw ReadError{"vector3 must contain numbers"};     // :442-448
    return r;
}

double get_positive_double(const Value& obj, const std::string& key) {
    double x = get_double(obj, key);
    if (x <= 0.0) throw ReadError{"key \"" + key + "\" must be positive"};         // :451-457
    return x;
}
double get_negative_double(const Value& obj, const std::string& key) {
    double x = get_double(obj, key);
    if (x >= 0.0) throw ReadError{"key \"" + key + "\" must be negative"};         // :459-465
    return x;
}
double get_nonnegative_double(const Value& obj, const std::string& key) {
    double x = get_double(obj, key);
    if (x < 0.0) throw ReadError{"key \"" + key + "\" must be non-negative"};      // :467-473
    return x;
}

Vec3 get_color(const Value& obj, const std::string& key) {
    Vec3 c = get_vector3(obj, key);                                                 // :475-487
    if (c.x < 0.0 || c.x > 1.0) throw ReadError{"color has r component outside the range [0, 1]"};
    if (c.y < 0.0 || c.y > 1.0) throw ReadError{"color has g component outside the range [0, 1]"};
    if (c.z < 0.0 || c.z > 1.0) throw ReadError{"color has b component outside the range [0, 1]"};
    return c;
}

struct Material { std::string name; double shininess; Vec3 color; };

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

void build(const Value& j, mrt_loaded_scene& out) {
    if (!j.is_object()) throw ReadError{"rayson must be comprised of one JSON object"};  // :382-384

    // camera, rayson.hpp:489-491
    Vec3 eye = get_vector3(j, "camera_eye");
    Vec3 up = get_vector3(j, "camera_up");
    Vec3 view = get_vector3(j, "camera_view");

    // viewport, :493-498
    unsigned xres = get_positive_unsigned(j, "x_resolution");
    unsigned yres = get_positive_unsigned(j, "y_resolution");
    double left = get_negative_double(j, "viewport_left");
    double top = get_positive_double(j, "viewport_top");
    double right = get_positive_double(j, "viewport_right");
    double bottom = get_negative_double(j, "viewport_bottom");

    // projection, :500-521
    int projection = MRT_PROJ_ORTHO;
    double focal = 0.0;
    {
        const std::string ortho_key = "ortho_projection", persp_key = "persp_focal_length";
        bool has_orth = has(j, ortho_key), has_persp = has(j, persp_key);
        if (has_orth && has_persp) throw ReadError{"cannot have both " + ortho_key + " and " + persp_key};
        if (!has_orth && !has_persp) throw ReadError{"must have " + ortho_key + " or " + persp_key};
        if (has_orth) {
            if (!get_bool(j, ortho_key)) throw ReadError{ortho_key + ", if present, must be true"};
            projection = MRT_PROJ_ORTHO;
        } else {
            projection = MRT_PROJ_PERSP;
            focal = get_positive_double(j, persp_key);
        }
    }

    // shader, :523-540 (flat_shader wins when both are present)
    int shader = MRT_SHADER_FLAT;
    double ka = 0, kd = 0, ks = 0;
    Vec3 amb;
    if (has(j, "flat_shader")) {
        if (!get_bool(j, "flat_shader")) throw ReadError{"flat_shader, if present, must be true"};
    } else if (has(j, "phong_shader")) {
        const Value& child = *j.find("phong_shader");
        shader = MRT_SHADER_PHONG;
        ka = get_nonnegative_double(child, "ambient_coeff");
        kd = get_nonnegative_double(child, "diffuse_coeff");
        ks = get_nonnegative_double(child, "specular_coeff");
        amb = get_color(child, "ambient_color");
    } else {
        throw ReadError{"must have flat_shader or phong_shader"};
    }

    Vec3 background = get_color(j, "background");                                   // :542

    // point lights (optional), :550-560
    struct Light { Vec3 loc, color; double intensity; };
    std::vector<Light> lights;
    if (has(j, "point_lights")) {
        const Value& child = *j.find("point_lights");
        if (!child.is_array()) throw ReadError{"expected point_lights to be an array"};
        for (const Value* it : child.children()) {
            Light l;
            l.loc = get_vector3(*it, "location");
            l.color = get_color(*it, "color");
            l.intensity = get_positive_double(*it, "intensity");
            lights.push_back(l);
        }
    }

    // materials (required), :562-581
    std::vector<Material> materials;
    if (!has(j, "materials")) throw ReadError{"no materials list"};
    for (const Value* it : j.find("materials")->children()) {
        Material m;
        m.name = get_string(*it, "name");
        m.shininess = get_positive_double(*it, "shininess");
        m.color = get_color(*it, "color");
        materials.push_back(std::move(m));
    }
    std::unordered_map<std::string, uint32_t> material_map;
    for (uint32_t k = 0; k < materials.size(); ++k) {
        if (material_map.count(materials[k].name) > 0)
            throw ReadError{"duplicate material name \"" + materials[k].name + "\""};
        material_map[materials[k].name] = k;
    }

    // spheres (optional), :583-594
    struct Sphere { uint32_t mat; Vec3 c; double r; };
    std::vector<Sphere> spheres;
    if (has(j, "spheres")) {
        for (const Value* it : j.find("spheres")->children()) {
            std::string name = get_string(*it, "material");
            auto found = material_map.find(name);
            if (found == material_map.end())
                throw ReadError{"sphere references undefined material \"" + name + "\""};
            Sphere s;
            s.mat = found->second;
            s.c = get_vector3(*it, "center");
            s.r = get_positive_double(*it, "radius");
            spheres.push_back(s);
        }
    }

    // triangles (optional), :596-615
    struct Tri { uint32_t mat; Vec3 a, b, c; };
    std::vector<Tri> tris;
    if (has(j, "triangles")) {
        for (const Value* it : j.find("triangles")->children()) {
            std::string name = get_string(*it, "material");
            Tri t;
            t.a = get_vector3(*it, "a");
            t.b = get_vector3(*it, "b");
            t.c = get_vector3(*it, "c");
            if (same(t.a, t.b) || same(t.a, t.c) || same(t.b, t.c))
                throw ReadError{"triangle is degenerate due to duplicated vertices"};
            auto found = material_map.find(name);
            if (found == material_map.end())
                throw ReadError{"triangle references undefined material \"" + name + "\""};
            t.mat = found->second;
            tris.push_back(t);
        }
    }

    // ---- conversion (gfxraytrace.hpp:645-721, :727-808, :881-889) ----
    mrt_scene& s = out.view;
    mrt_scene_info& info = out.info;

    const double vm = std::sqrt(view.x * view.x + view.y * view.y + view.z * view.z);
    const double w[3] = {-(view.x / vm), -(view.y / vm), -(view.z / vm)};
    const double cx[3] = {up.y * w[2] - up.z * w[1], up.z * w[0] - up.x * w[2], up.x * w[1] - up.y * w[0]};
    const double cm = std::sqrt(cx[0] * cx[0] + cx[1] * cx[1] + cx[2] * cx[2]);
    const double u[3] = {cx[0] / cm, cx[1] / cm, cx[2] / cm};
    const double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
    const double e[3] = {eye.x, eye.y, eye.z};
    for (int k = 0; k < 3; ++k) {
        info.eye[k] = e[k]; info.u[k] = u[k]; info.v[k] = v[k]; info.w[k] = w[k];
        s.eye[k] = static_cast<float>(e[k]);
        s.u[k] = static_cast<float>(u[k]);
        s.v[k] = static_cast<float>(v[k]);
        s.w[k] = static_cast<float>(w[k]);
    }
    s.xres = xres; s.yres = yres;
    info.left = left; info.top = top; info.right = right; info.bottom = bottom;
    s.left = static_cast<float>(left); s.right = static_cast<float>(right);
    s.top = static_cast<float>(top); s.bottom = static_cast<float>(bottom);
    s.projection = projection;
    info.focal = focal;
    s.focal = static_cast<float>(focal);
    s.shader = shader;
    info.ka = ka; info.kd = kd; info.ks = ks;
    s.ka = static_cast<float>(ka); s.kd = static_cast<float>(kd); s.ks = static_cast<float>(ks);
    s.ambient[0] = static_cast<float>(amb.x); s.ambient[1] = static_cast<float>(amb.y); s.ambient[2] = static_cast<float>(amb.z);
    s.background[0] = static_cast<float>(background.x);
    s.background[1] = static_cast<float>(background.y);
    s.background[2] = static_cast<float>(background.z);

    for (const auto& m : materials) {
        out.mat[0].push_back(static_cast<float>(m.color.x));
        out.mat[1].push_back(static_cast<float>(m.color.y));
        out.mat[2].push_back(static_cast<float>(m.color.z));
        out.mat[3].push_back(static_cast<float>(m.shininess));
        out.material_names.push_back(m.name);
    }
    for (const auto& sp : spheres) {
        out.sph[0].push_back(static_cast<float>(sp.c.x));
        out.sph[1].push_back(static_cast<float>(sp.c.y));
        out.sph[2].push_back(static_cast<float>(sp.c.z));
        out.sph[3].push_back(static_cast<float>(sp.r));
        out.sph_mat.push_back(sp.mat);
    }
    for (const auto& t : tris) {
        const double vals[9] = {t.a.x, t.a.y, t.a.z, t.b.x, t.b.y, t.b.z, t.c.x, t.c.y, t.c.z};
        for (int k = 0; k < 9; ++k) out.tri[k].push_back(static_cast<float>(vals[k]));
        out.tri_mat.push_back(t.mat);
    }
    for (const auto& l : lights) {
        out.light[0].push_back(static_cast<float>(l.loc.x));
        out.light[1].push_back(static_cast<float>(l.loc.y));
        out.light[2].push_back(static_cast<float>(l.loc.z));
        out.light[3].push_back(static_cast<float>(l.color.x));
        out.light[4].push_back(static_cast<float>(l.color.y));
        out.light[5].push_back(static_cast<float>(l.color.z));
        out.light_pos.push_back(l.loc.x); out.light_pos.push_back(l.loc.y); out.light_pos.push_back(l.loc.z);
        out.light_intensity.push_back(l.intensity);
    }

    s.n_spheres = static_cast<uint32_t>(spheres.size());
    for (int k = 0; k < 4; ++k) s.sph[k] = out.sph[k].data();
    s.sph_mat = out.sph_mat.data();
    s.n_tris = static_cast<uint32_t>(tris.size());
    for (int k = 0; k < 9; ++k) s.tri[k] = out.tri[k].data();
    s.tri_mat = out.tri_mat.data();
    s.n_materials = static_cast<uint32_t>(materials.size());
    for (int k = 0; k < 4; ++k) s.mat[k] = out.mat[k].data();
    s.n_lights = static_cast<uint32_t>(lights.size());
    for (int k = 0; k < 6; ++k) s.light[k] = out.light[k].data();
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
    } catch (const ReadError& e) {
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

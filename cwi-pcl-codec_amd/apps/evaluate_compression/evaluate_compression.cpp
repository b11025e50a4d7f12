// evaluate_compression.cpp -- the reference's evaluation app (apps/evaluate_compression, "eval.hpp" =
// include/pcl/apps/evaluate_compression/impl/evaluate_compression_impl.hpp) on top of the drop-in codec
// class: same options, same frame loop (group -> normalise -> encode -> decode -> quality -> CSV -> PLY),
// without Boost / PCL-io / VTK.  SURVEY.md section 8(f) row 1 ("harness parity").
//
// What is mirrored, with the reference lines:
//   options and their defaults            eval.hpp:135-171, command line first, then parameter_config.txt from the
//                                         parent or the current directory (eval.hpp:256-303)
//   codec construction                    eval.hpp:377-417 (through the shim class, same 14 arguments)
//   file list, sorted; .ply / .pcd        eval.hpp:649-694
//   grouping                              eval.hpp:754-783
//   per group                             eval.hpp:798-893: deep copy, normalize_pointclouds, per frame encode /
//                                         decode / computeQualityMetric / csv line / restore_scaling / .ply output
//   csv header and line                   quality_metrics_impl.hpp:242-285 (same stream formatting: operator<<)
//   delta (predictive) coding branch      eval.hpp:498-527, 854-890 (do_delta_coding, icp_on_original, predictive csv,
//                                         delta_decoded_pc_<n>.ply)
//   radius outlier filter                 eval.hpp:430-435 (K_outlier_filter, radius)
// Not part of this build: the V1 algorithm and the VTK windows; asking for them prints a note and goes on without.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <pcl/cloud_codec_v2/point_cloud_codec_v2.h>

typedef pcl::PointXYZRGB PointT;
typedef pcl::PointCloud<PointT> Cloud;
typedef Cloud::Ptr CloudPtr;
typedef pcl::io::OctreePointCloudCodecV2<PointT> Codec;

namespace {

// ------------------------------------------------------------------------------------------------
// options (eval.hpp:135-171)
// ------------------------------------------------------------------------------------------------
struct OptionDef { const char* name; char short_name; const char* def; bool is_bool; bool implicit_true; };
const OptionDef kOptions[] = {
    {"help", 'h', "", true, true},
    {"K_outlier_filter", 'K', "0", false, false},
    {"radius", 0, "0.01", false, false},
    {"group_size", 'g', "0", false, false},
    {"bb_expand_factor", 'f', "0.20", false, false},
    {"algorithm", 'a', "V2", false, false},
    {"input_directories", 'i', "", false, false},
    {"output_directory", 'o', "", false, false},
    {"show_statistics", 's', "0", true, true},
    {"visualization", 'v', "0", true, true},
    {"point_resolution", 'p', "0.20", false, false},
    {"octree_resolution", 'r', "0.20", false, false},
    {"octree_bits", 'b', "11", false, false},
    {"color_bits", 'c', "8", false, false},
    {"enh_bits", 'e', "0", false, false},
    {"color_coding_type", 't', "1", false, false},
    {"macroblock_size", 'm', "16", false, false},
    {"keep_centroid", 0, "0", false, false},
    {"create_scalable", 0, "0", true, false},
    {"do_connectivity_coding", 0, "0", true, false},
    {"icp_on_original", 0, "0", true, false},
    {"jpeg_quality", 'j', "0", false, false},
    {"do_delta_coding", 'd', "0", true, false},
    {"do_quality_computation", 'q', "0", true, false},
    {"do_icp_color_offset", 0, "0", true, false},
    {"num_threads", 'n', "1", false, false},
    {"intra_frame_quality_csv", 0, "intra_frame_quality.csv", false, false},
    {"predictive_quality_csv", 0, "predictive_quality.csv", false, false},
    {"debug_level", 0, "0", false, false},
    // not in the reference: load the input, print point counts and a checksum per file, and stop (no GPU needed)
    {"list_only", 0, "0", true, true},
    // not in the reference: GPUs that share the frames of a group, "0,1,2,..." (frame f on the f mod N-th of them through
    // pcc_pipeline_create_multi; bitstreams and frame ids as in the serial loop).  Empty: the reference's loop on PCC_DEVICE.
    {"devices", 0, "", false, false},
};

struct Options {
  std::map<std::string, std::string> value;       // explicit values
  std::vector<std::string> input_directories;
  const OptionDef* find(const std::string& name) const {
    for (const OptionDef& o : kOptions)
      if (name == o.name) return &o;
    return nullptr;
  }
  const OptionDef* find_short(char c) const {
    for (const OptionDef& o : kOptions)
      if (o.short_name && o.short_name == c) return &o;
    return nullptr;
  }
  std::string str(const char* name) const {
    auto it = value.find(name);
    if (it != value.end()) return it->second;
    return find(name)->def;
  }
  static bool truth(const std::string& s) { return s == "1" || s == "true" || s == "on" || s == "yes"; }
  bool flag(const char* name) const { return truth(str(name)); }
  int integer(const char* name) const { return atoi(str(name).c_str()); }
  double real(const char* name) const { return atof(str(name).c_str()); }
  void set(const OptionDef* o, const std::string& v, bool override_existing) {
    if (std::string(o->name) == "input_directories") {
      if (override_existing || input_directories.empty()) input_directories.push_back(v);
      return;
    }
    if (override_existing || !value.count(o->name)) value[o->name] = v;
  }
};

std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// command line first, then the optional parameter_config.txt: values given on the command line win (eval.hpp:283-300)
bool get_options(int argc, char** argv, Options& opt) {
  std::vector<std::string> unknown;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    const OptionDef* o = nullptr;
    std::string v;
    bool have_value = false;
    if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      const size_t eq = a.find('=');
      o = opt.find(a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2));
      if (eq != std::string::npos) { v = a.substr(eq + 1); have_value = true; }
    } else if (a.size() >= 2 && a[0] == '-' && !isdigit((unsigned char)a[1])) {
      o = opt.find_short(a[1]);
      if (a.size() > 2) { v = a.substr(2); have_value = true; }
    } else {  // positional = input directory (eval.hpp:172)
      opt.input_directories.push_back(a);
      continue;
    }
    if (!o) { unknown.push_back(a); continue; }
    if (!have_value) {
      if (o->implicit_true) {
        v = "1";
      } else if (i + 1 < argc) {
        v = argv[++i];
      } else {
        std::cerr << "option '" << a << "' needs a value\n";
        return false;
      }
    }
    opt.set(o, v, true);
  }
  if (!unknown.empty()) {
    std::cerr << "Unrecognized options on command line:\n";
    for (const std::string& u : unknown) std::cerr << u << "\n";
    return false;
  }
  std::ifstream cfg("../parameter_config.txt");
  if (cfg.fail()) {
    cfg.clear();
    cfg.open("parameter_config.txt");
    if (cfg.fail()) {
      std::cerr << " Optional file 'parameter_config.txt' not found in the current directory or its parent.\n";
      return true;
    }
  }
  std::string line;
  bool ok = true;
  while (std::getline(cfg, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
    const OptionDef* o = opt.find(key);
    if (!o) {
      if (ok) std::cerr << "Unrecognized options in configuration file:\n";
      std::cerr << key << "\n";
      ok = false;
      continue;
    }
    opt.set(o, val, false);
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------
// files (eval.hpp:607-694)
// ------------------------------------------------------------------------------------------------
bool ends_with(const std::string& s, const char* tail) {
  const size_t n = strlen(tail);
  return s.size() >= n && s.compare(s.size() - n, n, tail) == 0;
}

int get_filenames_from_dir(const std::string& directory, std::vector<std::string>& filenames) {
  struct stat st;
  if (stat(directory.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) {
    std::cerr << "'" << directory << "' is not a directory.\n";
    return -1;
  }
  DIR* d = opendir(directory.c_str());
  if (!d) return -1;
  while (dirent* e = readdir(d)) {
    const std::string name = e->d_name;
    if (name == "." || name == "..") continue;
    filenames.push_back(directory + (ends_with(directory, "/") ? "" : "/") + name);
  }
  closedir(d);
  std::sort(filenames.begin(), filenames.end());
  return 0;
}

size_t type_size(const std::string& t) {
  if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
  if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
  if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
  if (t == "double" || t == "float64" || t == "int64" || t == "uint64") return 8;
  return 0;
}
double read_scalar(const unsigned char* p, const std::string& t) {
  if (t == "float" || t == "float32") { float v; memcpy(&v, p, 4); return v; }
  if (t == "double" || t == "float64") { double v; memcpy(&v, p, 8); return v; }
  if (t == "uchar" || t == "uint8") return *p;
  if (t == "char" || t == "int8") return *reinterpret_cast<const signed char*>(p);
  if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, p, 2); return v; }
  if (t == "short" || t == "int16") { int16_t v; memcpy(&v, p, 2); return v; }
  if (t == "uint" || t == "uint32") { uint32_t v; memcpy(&v, p, 4); return v; }
  if (t == "int" || t == "int32") { int32_t v; memcpy(&v, p, 4); return v; }
  return 0.0;
}

// PLY: ascii and binary_little_endian, element "vertex" with x y z (+ red green blue [alpha]); other elements ignored
bool load_ply_file(const std::string& path, Cloud& pc) {
  std::ifstream in(path.c_str(), std::ios::binary);
  if (!in) return false;
  std::string line;
  if (!std::getline(in, line) || trim(line) != "ply") return false;
  bool binary = false;
  size_t n_vertex = 0;
  struct Prop { std::string type, name; bool is_list; };
  std::vector<Prop> props;
  bool in_vertex = false, vertex_first = true, seen_other = false;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string w;
    ss >> w;
    if (w == "format") {
      ss >> w;
      if (w == "binary_little_endian") binary = true;
      else if (w != "ascii") return false;
    } else if (w == "element") {
      std::string name;
      size_t count = 0;
      ss >> name >> count;
      in_vertex = name == "vertex";
      if (in_vertex) { n_vertex = count; vertex_first = !seen_other; }
      else seen_other = true;
    } else if (w == "property" && in_vertex) {
      Prop p;
      ss >> p.type;
      p.is_list = p.type == "list";
      if (p.is_list) { std::string a, b; ss >> a >> b; }
      ss >> p.name;
      props.push_back(p);
    } else if (w == "end_header") {
      break;
    }
  }
  if (!vertex_first) return false;  // vertex data behind another element: not produced by the tools this app reads from
  pc.points.assign(n_vertex, PointT());
  size_t stride = 0;
  for (const Prop& p : props) { if (p.is_list) return false; stride += type_size(p.type); }
  std::vector<unsigned char> row(stride);
  for (size_t i = 0; i < n_vertex; ++i) {
    PointT& q = pc.points[i];
    double vals[64];
    if (binary) {
      in.read(reinterpret_cast<char*>(row.data()), (std::streamsize)stride);
      if (!in) return false;
      size_t off = 0;
      for (size_t k = 0; k < props.size() && k < 64; ++k) { vals[k] = read_scalar(row.data() + off, props[k].type); off += type_size(props[k].type); }
    } else {
      for (size_t k = 0; k < props.size() && k < 64; ++k)
        if (!(in >> vals[k])) return false;
    }
    for (size_t k = 0; k < props.size() && k < 64; ++k) {
      const std::string& nm = props[k].name;
      if (nm == "x") q.x = (float)vals[k];
      else if (nm == "y") q.y = (float)vals[k];
      else if (nm == "z") q.z = (float)vals[k];
      else if (nm == "red" || nm == "r") q.r = (uint8_t)vals[k];
      else if (nm == "green" || nm == "g") q.g = (uint8_t)vals[k];
      else if (nm == "blue" || nm == "b") q.b = (uint8_t)vals[k];
      else if (nm == "alpha") q.a = (uint8_t)vals[k];
    }
  }
  pc.width = (uint32_t)n_vertex;
  pc.height = 1;
  return true;
}

// PCD: FIELDS x y z [rgb|rgba], DATA ascii or binary
bool load_pcd_file(const std::string& path, Cloud& pc) {
  std::ifstream in(path.c_str(), std::ios::binary);
  if (!in) return false;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  size_t n_points = 0;
  std::string data, line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string w;
    ss >> w;
    if (w == "FIELDS") { std::string f; while (ss >> f) fields.push_back(f); }
    else if (w == "SIZE") { int v; while (ss >> v) sizes.push_back(v); }
    else if (w == "TYPE") { std::string t; while (ss >> t) types.push_back(t); }
    else if (w == "COUNT") { int v; while (ss >> v) counts.push_back(v); }
    else if (w == "POINTS") ss >> n_points;
    else if (w == "DATA") { ss >> data; break; }
  }
  if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size()) return false;
  if (counts.size() != fields.size()) counts.assign(fields.size(), 1);
  if (data != "ascii" && data != "binary") {
    std::cerr << path << ": PCD DATA " << data << " is not supported\n";
    return false;
  }
  pc.points.assign(n_points, PointT());
  size_t stride = 0;
  for (size_t k = 0; k < fields.size(); ++k) stride += (size_t)sizes[k] * (size_t)counts[k];
  std::vector<unsigned char> row(stride);
  for (size_t i = 0; i < n_points; ++i) {
    PointT& q = pc.points[i];
    size_t off = 0;
    if (data == "binary") {
      in.read(reinterpret_cast<char*>(row.data()), (std::streamsize)stride);
      if (!in) return false;
    }
    for (size_t k = 0; k < fields.size(); ++k) {
      for (int c = 0; c < counts[k]; ++c) {
        float fv = 0.f;
        uint32_t uv = 0;
        if (data == "binary") {
          if (types[k] == "F" && sizes[k] == 4) { memcpy(&fv, row.data() + off, 4); memcpy(&uv, row.data() + off, 4); }
          else if (types[k] == "F" && sizes[k] == 8) { double d; memcpy(&d, row.data() + off, 8); fv = (float)d; }
          else if (sizes[k] == 4) { memcpy(&uv, row.data() + off, 4); fv = (float)uv; }
          off += (size_t)sizes[k];
        } else {
          std::string tok;
          if (!(in >> tok)) return false;
          if (types[k] == "F") { fv = (float)atof(tok.c_str()); memcpy(&uv, &fv, 4); }
          else { uv = (uint32_t)strtoul(tok.c_str(), nullptr, 10); fv = (float)uv; }
        }
        if (c) continue;
        if (fields[k] == "x") q.x = fv;
        else if (fields[k] == "y") q.y = fv;
        else if (fields[k] == "z") q.z = fv;
        else if (fields[k] == "rgb" || fields[k] == "rgba") { q.rgba = uv; if (fields[k] == "rgb") q.a = 255; }
      }
    }
  }
  pc.width = (uint32_t)n_points;
  pc.height = 1;
  return true;
}

bool load_input_cloud(const std::string& filename, Cloud& pc) {
  if (ends_with(filename, ".ply")) return load_ply_file(filename, pc);
  if (ends_with(filename, ".pcd")) return load_pcd_file(filename, pc);
  return false;
}

// .ply output of the decoded cloud (eval.hpp:532-540: PLYWriter on a PCLPointCloud2, ascii)
void write_ply(const std::string& path, const Cloud& pc) {
  std::ofstream out(path.c_str());
  out << "ply\nformat ascii 1.0\ncomment written by pcc evaluate_compression\nelement vertex " << pc.points.size()
      << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
      << "end_header\n";
  for (const PointT& p : pc.points) out << p.x << " " << p.y << " " << p.z << " " << (int)p.r << " " << (int)p.g << " " << (int)p.b << "\n";
}

// ------------------------------------------------------------------------------------------------
// QualityMetric (quality_metrics.h:53-80, quality_metrics_impl.hpp:242-285)
// ------------------------------------------------------------------------------------------------
struct QualityMetric {
  size_t compressed_size = 0;
  uint64_t in_point_count = 0, out_point_count = 0, byte_count_octree_layer = 0, byte_count_centroid_layer = 0, byte_count_color_layer = 0;
  float symm_rms = 0, symm_hausdorff = 0, left_hausdorff = 0, right_hausdorff = 0, left_rms = 0, right_rms = 0;
  double psnr_db = 0, psnr_yuv[3] = {0, 0, 0};
  double encoding_time_ms = 0, decoding_time_ms = 0;

  static void print_csv_header(std::ostream& csv) {
    csv << "compression setting; "
        << "in point count;" << "out point count;" << "compressed_byte_size;" << "compressed_byte_size_per_output_point;"
        << "octree_byte_size_per_voxel;" << "centroid_byte_size_per_voxel;" << "color_byte_size_per_voxel;" << "symm_rms;"
        << "symm_haussdorff;" << "psnr_db;" << "psnr_colors_y;" << "psnr_colors_u;" << "psnr_colors_v;" << "encoding_time_ms;"
        << "decoding_time_ms;" << std::endl;
  }
  void print_csv_line(const std::string& setting, std::ostream& csv) const {
    csv << setting << ";" << in_point_count << ";" << out_point_count << ";" << compressed_size << ";"
        << compressed_size / (1.0 * out_point_count) << ";" << byte_count_octree_layer / (1.0 * out_point_count) << ";"
        << byte_count_centroid_layer / (1.0 * out_point_count) << ";" << byte_count_color_layer / (1.0 * out_point_count) << ";"
        << symm_rms << ";" << symm_hausdorff << ";" << psnr_db << ";" << psnr_yuv[0] << ";" << psnr_yuv[1] << ";" << psnr_yuv[2] << ";"
        << encoding_time_ms << ";" << decoding_time_ms << ";" << std::endl;
  }
};

double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ------------------------------------------------------------------------------------------------
// the app (eval.hpp:341-428, 696-893)
// ------------------------------------------------------------------------------------------------
struct App {
  Options opt;
  std::unique_ptr<Codec> encoder, decoder;
  pcc_ctx* quality_ctx = nullptr;
  pcc_multi_pipeline* multi = nullptr;  // --devices: the GPUs that share the frames of a group
  int output_index = -1;
  std::ofstream predictive_csv;

  ~App() {
    if (quality_ctx) pcc_destroy(quality_ctx);
    if (multi) pcc_multi_pipeline_destroy(multi);
  }

  bool open_devices() {  // --devices 0,1,2,...
    const std::string list = opt.str("devices");
    if (list.empty() || multi) return true;
    std::vector<int> devs;
    std::stringstream ls(list);
    std::string item;
    while (std::getline(ls, item, ','))
      if (!item.empty()) devs.push_back(atoi(item.c_str()));
    if (devs.empty()) return true;
    const int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
    multi = pcc_pipeline_create_multi(devs.data(), (int)devs.size(), std::max(2, std::min(16, cpus / (int)devs.size())));
    if (!multi) std::cerr << "cannot open the GPUs of --devices " << list << ": " << pcc_multi_pipeline_last_error(nullptr) << "\n";
    return multi != nullptr;
  }

  void complete_initialization() {  // eval.hpp:341-428 (V2 branch)
    const int octree_bits = opt.integer("octree_bits"), enh_bits = opt.integer("enh_bits"), color_bits = opt.integer("color_bits");
    double point_resolution = opt.real("point_resolution"), octree_resolution = opt.real("octree_resolution");
    if (octree_bits > 0) {  // eval.hpp:381-384
      point_resolution = std::pow(2.0, -1.0 * (octree_bits + enh_bits));
      octree_resolution = std::pow(2.0, -1.0 * octree_bits);
    }
    auto make = [&]() {
      return new Codec(pcl::io::MANUAL_CONFIGURATION, opt.flag("show_statistics"), point_resolution, octree_resolution, true, 0,
                       color_bits ? true : false, (unsigned char)color_bits, (unsigned char)opt.integer("color_coding_type"),
                       opt.integer("keep_centroid") != 0, opt.flag("create_scalable"), false, opt.integer("jpeg_quality"),
                       opt.integer("num_threads"));
    };
    encoder.reset(make());
    decoder.reset(make());
    encoder->setMacroblockSize(opt.integer("macroblock_size"));
    encoder->setDoICPColorOffset(opt.flag("do_icp_color_offset"));
  }

  // do_quality_computation (eval.hpp:529-541): computeQualityMetric on the GPU
  void do_quality_computation(const CloudPtr& reference, const CloudPtr& cloud, QualityMetric& q, double res) {
    if (!quality_ctx) quality_ctx = pcc_create(pcl::io::pcc_shim_device());
    pcc_quality m;
    const int rc = quality_ctx ? pcc_quality_metrics(quality_ctx, reinterpret_cast<const pcc_point_xyzrgb*>(reference->points.data()), reference->points.size(),
                                                     reinterpret_cast<const pcc_point_xyzrgb*>(cloud->points.data()), cloud->points.size(), res, &m)
                               : PCC_ERR_HIP;
    if (rc == PCC_OK) {
      q.in_point_count = m.in_point_count; q.out_point_count = m.out_point_count;
      q.symm_rms = m.symm_rms; q.symm_hausdorff = m.symm_hausdorff; q.left_hausdorff = m.left_hausdorff;
      q.right_hausdorff = m.right_hausdorff; q.left_rms = m.left_rms; q.right_rms = m.right_rms; q.psnr_db = m.psnr_db;
      for (int c = 0; c < 3; ++c) q.psnr_yuv[c] = m.psnr_yuv[c];
      std::cout << "Symmetric Geometric Hausdorff Distance: " << q.symm_hausdorff << "\nSymmetric Geometric Root Mean Square Distance: "
                << q.symm_rms << "\nGeometric PSNR: " << q.psnr_db << " dB\nA->B color psnr Y: " << (float)q.psnr_yuv[0] << " dB U: "
                << (float)q.psnr_yuv[1] << " dB V: " << (float)q.psnr_yuv[2] << " dB\n";
    } else {
      std::cerr << "quality computation failed: " << (quality_ctx ? pcc_last_error(quality_ctx) : "no GPU") << "\n";
    }
  }

  bool evaluate_group(std::vector<CloudPtr>& group, const std::string& settings, std::ofstream& intra_csv) {
    std::vector<CloudPtr> working_group;
    for (CloudPtr& c : group) working_group.push_back(CloudPtr(new Cloud(*c)));  // deep copy (eval.hpp:804-808)
    if (opt.integer("K_outlier_filter") > 0)  // do_outlier_removal (eval.hpp:430-435)
      Codec::remove_outliers(working_group, opt.integer("K_outlier_filter"), opt.real("radius"), (unsigned)opt.integer("debug_level"));
    pcl::io::BoundingBox bb;  // do_bounding_box_normalization (eval.hpp:438-444), with the reference's vector type
    std::vector<float> dyn_range, offset;
    std::vector<pcl::io::BoundingBox, Eigen::aligned_allocator<pcl::io::BoundingBox> > boxes(working_group.size());
    const double f = opt.real("bb_expand_factor");
    if (f > 0.0) bb = Codec::normalize_pointclouds(working_group, boxes, f, dyn_range, offset, (unsigned)opt.integer("debug_level"));
    const double res = opt.integer("octree_bits") > 0 ? std::pow(2.0, -1.0 * opt.integer("octree_bits")) : opt.real("octree_resolution");
    // Several GPUs: the whole group is encoded up front, frame f on GPU f mod N (the frames are independent I-frames);
    // the loop below then takes each frame's bitstream instead of calling the encoder.  Delta coding needs the
    // encoder object's simplified cloud of every frame, so it keeps the reference's loop.
    std::vector<pcc_bitstream> pre;
    double pre_ms_per_frame = 0.0;
    if (multi && !opt.flag("do_delta_coding")) {
      std::vector<const void*> ptrs;
      std::vector<size_t> counts;
      for (CloudPtr& c : working_group) { ptrs.push_back(c->points.data()); counts.push_back(c->points.size()); }
      pre.assign(working_group.size(), pcc_bitstream());
      pcc_params prm = encoder->native_params();
      prm.frame_id = encoder->next_frame_id();
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = pcc_multi_pipeline_encode_host(multi, ptrs.data(), counts.data(), ptrs.size(), sizeof(PointT), 16, &prm, pre.data());
      pre_ms_per_frame = ms_since(t0) / (double)std::max<size_t>(1, ptrs.size());
      if (rc != PCC_OK) {
        std::cerr << "multi-GPU encode failed: " << pcc_multi_pipeline_last_error(multi) << "\n";
        return false;
      }
      size_t coded = 0;
      for (const pcc_bitstream& b : pre) coded += b.len ? 1 : 0;
      encoder->advance_frame_id((uint32_t)coded);
    }
    for (size_t i = 0; i < working_group.size(); ++i) {
      CloudPtr pc = working_group[i];
      QualityMetric q;
      std::stringstream ss;
      if (!pre.empty()) {
        ss.write(reinterpret_cast<const char*>(pre[i].data), (std::streamsize)pre[i].len);
        q.encoding_time_ms = pre_ms_per_frame;
        q.byte_count_octree_layer = pre[i].perf[0]; q.byte_count_centroid_layer = pre[i].perf[1]; q.byte_count_color_layer = pre[i].perf[2];
        q.compressed_size = pre[i].len;
        std::cout << " octreeCoding " << q.compressed_size << " bytes  base layer  " << std::endl;
      } else {  // do_encoding (eval.hpp:446-474)
        const auto t0 = std::chrono::steady_clock::now();
        encoder->encodePointCloud(pc, ss);
        q.encoding_time_ms = ms_since(t0);
        const uint64_t* c_sizes = encoder->getPerformanceMetrics();
        q.byte_count_octree_layer = c_sizes[0]; q.byte_count_centroid_layer = c_sizes[1]; q.byte_count_color_layer = c_sizes[2];
        q.compressed_size = (size_t)ss.tellp();
        std::cout << " octreeCoding " << q.compressed_size << " bytes  base layer  " << std::endl;
      }
      std::stringstream coded_stream(ss.str());
      CloudPtr output(new Cloud());
      {  // do_decoding (eval.hpp:476-494)
        const auto t0 = std::chrono::steady_clock::now();
        decoder->decodePointCloud(coded_stream, output);
        q.decoding_time_ms = ms_since(t0);
      }
      if (opt.flag("do_quality_computation")) {  // eval.hpp:836-843; computed on the normalised clouds
        do_quality_computation(pc, output, q, res);
        if (!opt.str("intra_frame_quality_csv").empty()) q.print_csv_line(settings, intra_csv);
      }
      CloudPtr rescaled(new Cloud(*output));
      if (f > 0.0) Codec::restore_scaling(rescaled, bb);  // eval.hpp:846
      if (!opt.str("output_directory").empty()) {
        std::ostringstream name;
        name << opt.str("output_directory") << "/pointcloud_" << output_index++ << ".ply";
        write_ply(name.str(), *rescaled);
      }
      // test and evaluation of the iterative-closest-point predictive coding (eval.hpp:853-890)
      if (opt.flag("do_delta_coding") && f >= 0.0 && i + 1 < working_group.size()) {
        CloudPtr predicted(new Cloud());
        std::cout << " delta coding frame nr " << i + 1 << std::endl;
        std::stringstream p_frame_pdat, p_frame_idat;
        QualityMetric pq;
        const bool icp_on_original = opt.flag("icp_on_original");
        {  // do_delta_encoding (eval.hpp:498-514)
          const auto t0 = std::chrono::steady_clock::now();
          encoder->encodePointCloudDeltaFrame(icp_on_original ? pc : encoder->getOutputCloud(), working_group[i + 1], predicted, p_frame_idat, p_frame_pdat,
                                              icp_on_original, false);
          pq.encoding_time_ms = ms_since(t0);
          pq.byte_count_octree_layer = (size_t)p_frame_idat.tellp();
          pq.byte_count_centroid_layer = (size_t)p_frame_pdat.tellp();
          pq.compressed_size = pq.byte_count_octree_layer + pq.byte_count_centroid_layer;
          pq.byte_count_color_layer = 0;
        }
        std::cout << " shared macroblocks " << encoder->getMacroBlockPercentage() << " of which icp converged "
                  << encoder->getMacroBlockConvergencePercentage() << std::endl;
        std::cout << " encoded a predictive frame: coded " << pq.byte_count_octree_layer << " bytes intra and " << pq.byte_count_centroid_layer
                  << " inter frame encoded " << std::endl;
        {  // do_delta_decoding (eval.hpp:516-527): from the decoded I frame
          const auto t0 = std::chrono::steady_clock::now();
          encoder->decodePointCloudDeltaFrame(output, predicted, p_frame_idat, p_frame_pdat);
          pq.decoding_time_ms = ms_since(t0);
        }
        if (opt.flag("do_quality_computation")) {
          do_quality_computation(working_group[i + 1], predicted, pq, res);
          if (predictive_csv.is_open()) pq.print_csv_line(settings, predictive_csv);
        }
        if (f > 0.0) Codec::restore_scaling(predicted, bb);
        if (!opt.str("output_directory").empty()) {
          std::ostringstream name;
          name << opt.str("output_directory") << "/delta_decoded_pc_" << output_index << ".ply";
          write_ply(name.str(), *predicted);
        }
      }
    }
    return true;
  }

  int run(int argc, char** argv) {
    if (!get_options(argc, argv, opt)) return 1;
    if (opt.value.count("help")) {
      std::cout << "options (defaults):\n";
      for (const OptionDef& o : kOptions) std::cout << "  --" << o.name << (o.short_name ? std::string(" [-") + o.short_name + "]" : std::string()) << " (" << o.def << ")\n";
      return 0;
    }
    if (opt.str("algorithm") != "V2") { std::cerr << "only algorithm V2 is part of this build\n"; return 1; }
    if (opt.flag("visualization")) std::cerr << "No visualization configured\n";
    if (opt.input_directories.size() > 1) { std::cout << "Fusing multiple directories not implemented.\n"; return 1; }
    if (opt.input_directories.empty()) { std::cout << "Need to specify a directory containing Point Cloud files (.pcd or .ply).\n"; return 1; }
    std::vector<std::string> filenames;
    if (get_filenames_from_dir(opt.input_directories[0], filenames) != 0) return 1;

    if (opt.flag("list_only")) {
      for (const std::string& fn : filenames) {
        Cloud pc;
        if (!load_input_cloud(fn, pc)) continue;
        uint64_t h = 1469598103934665603ull;
        for (const PointT& p : pc.points) {
          uint32_t w[4];
          memcpy(w, &p.x, 12);
          w[3] = p.rgba & 0x00ffffffu;
          for (int k = 0; k < 4; ++k) { h ^= w[k]; h *= 1099511628211ull; }
        }
        std::cout << fn << " " << pc.points.size() << " " << h << "\n";
      }
      return 0;
    }

    complete_initialization();
    if (!open_devices()) return 1;
    std::ostringstream settings;  // eval.hpp:741
    settings << "octree_bits=" << opt.integer("octree_bits") << " color_bits=" << opt.integer("color_bits") << " enh._bits=" << opt.integer("enh_bits")
             << "_colortype=" << opt.integer("color_coding_type") << " centroid=" << opt.integer("keep_centroid");
    std::ofstream intra_csv;
    if (!opt.str("intra_frame_quality_csv").empty()) {
      intra_csv.open(opt.str("intra_frame_quality_csv").c_str());
      QualityMetric::print_csv_header(intra_csv);
    }
    if (!opt.str("predictive_quality_csv").empty()) {
      predictive_csv.open(opt.str("predictive_quality_csv").c_str());
      QualityMetric::print_csv_header(predictive_csv);
    }
    const int group_size = opt.integer("group_size");
    std::vector<CloudPtr> group;
    size_t count = 0;
    for (const std::string& filename : filenames) {
      if (output_index == -1) {  // index of the first file, if its name starts with one (eval.hpp:757-763)
        std::stringstream ss(filename);
        ss >> output_index;
        if (ss.fail() || output_index == -1) output_index = 0;
      }
      CloudPtr pc(new Cloud());
      if (!load_input_cloud(filename, *pc)) continue;
      group.push_back(pc);
      ++count;
      if (group_size == 0 && count < filenames.size()) continue;
      if (group_size == 0 || count == filenames.size() || count % (size_t)group_size == 0) {
        evaluate_group(group, settings.str(), intra_csv);
        complete_initialization();  // a new encoder / decoder per group: frame ids restart (eval.hpp:779)
        group.clear();
        count = 0;
      }
    }
    if (!group.empty()) {  // files that failed to load left the last group open
      evaluate_group(group, settings.str(), intra_csv);
    }
    return 0;
  }
};

}  // namespace

int main(int argc, char** argv) {
  try {
    App app;
    return app.run(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << "evaluate_compression: " << e.what() << "\n";
    return 1;
  }
}

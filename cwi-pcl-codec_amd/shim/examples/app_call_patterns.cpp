// app_call_patterns.cpp -- the reference evaluation app's calls into the codec class, statement for statement, against
// the drop-in header.  Each block cites the lines of
//   apps/evaluate_compression/include/pcl/apps/evaluate_compression/impl/evaluate_compression_impl.hpp  ("eval.hpp")
// whose call it repeats with the same argument types (boost::shared_ptr clouds, the Eigen::aligned_allocator vector of
// boxes, std::stringstream pointers, bool / int / float setters).  It is compiled with plain g++ by
// tests/test_shim_boundary.py (no GPU needed to compile and link); run on a GPU box it executes the sequence once
// and checks the round trip.
//
//   g++ -std=c++11 -I cwi-pcl-codec_amd/shim -I include app_call_patterns.cpp -L cwi-pcl-codec_amd -lpcc_hip
#include <pcl/cloud_codec_v2/point_cloud_codec_v2.h>

#include <cmath>
#include <cstdio>
#include <sstream>
#include <string>
#include <vector>

using namespace std;  // the app's headers do the same (eval.hpp:442 writes `vector<...>` unqualified)

template <typename PointT>
struct app_like {
  // eval.hpp:124-125
  boost::shared_ptr<pcl::io::OctreePointCloudCodecV2<PointT> > encoder_V2_;
  boost::shared_ptr<pcl::io::OctreePointCloudCodecV2<PointT> > decoder_V2_;
  bool show_statistics_ = false, keep_centroid_ = false, create_scalable_ = false, do_icp_color_offset_ = false, icp_on_original_ = false;
  int octree_bits_ = 8, enh_bits_ = 0, color_bits_ = 8, color_coding_type_ = 1, jpeg_quality_ = 85, num_threads_ = 1, macroblock_size_ = 16;
  int K_outlier_filter_ = 0;
  double point_resolution_ = 0.2, bb_expand_factor_ = 0.2, radius_ = 0.01;
  unsigned int debug_level_ = 0;

  void complete_initialization() {
    // eval.hpp:377-395
    encoder_V2_ = boost::shared_ptr<pcl::io::OctreePointCloudCodecV2<PointT> >(
        new pcl::io::OctreePointCloudCodecV2<PointT>(
            pcl::io::MANUAL_CONFIGURATION, show_statistics_,
            octree_bits_ > 0 ? std::pow(2.0, -1.0 * (octree_bits_ + enh_bits_)) : point_resolution_,
            octree_bits_ > 0 ? std::pow(2.0, -1.0 * octree_bits_) : point_resolution_,
            true, 0, color_bits_ > 0 ? true : false, color_bits_, color_coding_type_, keep_centroid_, create_scalable_, false,
            jpeg_quality_, num_threads_));
    // eval.hpp:396-414
    decoder_V2_ = boost::shared_ptr<pcl::io::OctreePointCloudCodecV2<PointT> >(
        new pcl::io::OctreePointCloudCodecV2<PointT>(
            pcl::io::MANUAL_CONFIGURATION, false,
            octree_bits_ > 0 ? std::pow(2.0, -1.0 * (octree_bits_ + enh_bits_)) : point_resolution_,
            octree_bits_ > 0 ? std::pow(2.0, -1.0 * octree_bits_) : point_resolution_,
            true, 0, color_bits_ > 0 ? true : false, color_bits_, color_coding_type_, keep_centroid_, create_scalable_, false,
            jpeg_quality_, num_threads_));
    // eval.hpp:415-417
    encoder_V2_->setMacroblockSize(macroblock_size_);
    encoder_V2_->setDoICPColorOffset(do_icp_color_offset_);
  }

  // eval.hpp:432-435
  void do_outlier_removal(std::vector<boost::shared_ptr<pcl::PointCloud<PointT> > >& group) {
    pcl::io::OctreePointCloudCodecV2<PointT>::remove_outliers(group, K_outlier_filter_, radius_, debug_level_);
  }

  // eval.hpp:438-444
  pcl::io::BoundingBox do_bounding_box_normalization(std::vector<boost::shared_ptr<pcl::PointCloud<PointT> > >& group) {
    vector<float> dyn_range, offset;
    vector<pcl::io::BoundingBox, Eigen::aligned_allocator<pcl::io::BoundingBox> > bounding_boxes(group.size());
    return pcl::io::OctreePointCloudCodecV2<PointT>::normalize_pointclouds(group, bounding_boxes, bb_expand_factor_, dyn_range, offset,
                                                                           debug_level_);
  }

  // eval.hpp:461-470
  void do_encoding(boost::shared_ptr<pcl::PointCloud<PointT> > pointcloud, std::stringstream* stream, uint64_t sizes[3]) {
    encoder_V2_->encodePointCloud(pointcloud, *stream);
    std::uint64_t* c_sizes = encoder_V2_->getPerformanceMetrics();
    sizes[0] = c_sizes[0]; sizes[1] = c_sizes[1]; sizes[2] = c_sizes[2];
  }

  // eval.hpp:488-491
  void do_decoding(std::stringstream* coded_stream, boost::shared_ptr<pcl::PointCloud<PointT> > pointcloud) {
    decoder_V2_->decodePointCloud(*coded_stream, pointcloud);
  }

  // eval.hpp:498-513
  void do_delta_encoding(boost::shared_ptr<pcl::PointCloud<PointT> > i_cloud, boost::shared_ptr<pcl::PointCloud<PointT> > p_cloud,
                         boost::shared_ptr<pcl::PointCloud<PointT> > out_cloud, std::stringstream* i_stream, std::stringstream* p_stream) {
    encoder_V2_->encodePointCloudDeltaFrame(i_cloud, p_cloud, out_cloud, *i_stream, *p_stream, icp_on_original_, false);
  }

  // eval.hpp:516-527 (the app decodes predicted frames with the ENCODER object)
  void do_delta_decoding(std::stringstream* i_stream, std::stringstream* p_stream, boost::shared_ptr<pcl::PointCloud<PointT> > i_cloud,
                         boost::shared_ptr<pcl::PointCloud<PointT> > out_cloud) {
    encoder_V2_->decodePointCloudDeltaFrame(i_cloud, out_cloud, *i_stream, *p_stream);
  }

  // eval.hpp:800-893, without metrics / output / visualisation
  int evaluate_group(std::vector<boost::shared_ptr<pcl::PointCloud<PointT> > >& group, bool do_delta_coding_) {
    std::vector<boost::shared_ptr<pcl::PointCloud<PointT> > > working_group;
    for (typename std::vector<boost::shared_ptr<pcl::PointCloud<PointT> > >::iterator itr = group.begin(); itr != group.end(); itr++) {
      boost::shared_ptr<pcl::PointCloud<PointT> > point_cloud = *itr;
      working_group.push_back(point_cloud->makeShared());
    }
    if (K_outlier_filter_ > 0) do_outlier_removal(working_group);
    pcl::io::BoundingBox bb;
    if (bb_expand_factor_ > 0.0) bb = do_bounding_box_normalization(working_group);
    int bad = 0;
    for (int i = 0; i < (int)working_group.size(); i++) {
      boost::shared_ptr<pcl::PointCloud<PointT> > pc = working_group[i];
      stringstream ss;
      uint64_t sizes[3];
      do_encoding(pc, &ss, sizes);
      string s = ss.str();
      std::stringstream coded_stream(s);
      size_t group_size = group.size();
      boost::shared_ptr<pcl::PointCloud<PointT> > output_pointcloud(new pcl::PointCloud<PointT>());
      do_decoding(&coded_stream, output_pointcloud);
      boost::shared_ptr<pcl::PointCloud<PointT> > rescaled_pc = output_pointcloud->makeShared();
      if (bb_expand_factor_ > 0.0) pcl::io::OctreePointCloudCodecV2<PointT>::restore_scaling(rescaled_pc, bb);
      if (output_pointcloud->size() != encoder_V2_->getOutputCloud()->size()) ++bad;
      std::printf("frame %d: %zu points -> %zu bytes (octree %llu, centroid %llu, colour %llu) -> %zu voxels\n", i, pc->size(), s.size(),
                  (unsigned long long)sizes[0], (unsigned long long)sizes[1], (unsigned long long)sizes[2], output_pointcloud->size());
      if (do_delta_coding_ && bb_expand_factor_ >= 0 && i + 1 < (int)group_size) {  // eval.hpp:853-890
        boost::shared_ptr<pcl::PointCloud<PointT> > predicted_pc(new pcl::PointCloud<PointT>());
        stringstream p_frame_pdat, p_frame_idat;
        do_delta_encoding(icp_on_original_ ? pc : encoder_V2_->getOutputCloud(), working_group[i + 1], predicted_pc, &p_frame_idat, &p_frame_pdat);
        const float shared = encoder_V2_->getMacroBlockPercentage();
        const float converged = encoder_V2_->getMacroBlockConvergencePercentage();
        const int i_bytes = (int)p_frame_idat.tellp(), p_bytes = (int)p_frame_pdat.tellp();
        do_delta_decoding(&p_frame_idat, &p_frame_pdat, output_pointcloud, predicted_pc);
        pcl::io::OctreePointCloudCodecV2<PointT>::restore_scaling(predicted_pc, bb);
        std::printf("  predicted frame %d: %d bytes intra + %d bytes inter, %.3f shared, %.3f converged, %zu points\n", i + 1, i_bytes, p_bytes,
                    shared, converged, predicted_pc->size());
        if (predicted_pc->size() == 0) ++bad;
      }
      output_pointcloud->clear();
    }
    return bad;
  }
};

// the remaining public setters and the serial delta coder (codec.h:149-183): every overload must resolve as in the reference
template <typename PointT>
void touch_whole_surface(pcl::io::OctreePointCloudCodecV2<PointT>& c, const typename pcl::PointCloud<PointT>::ConstPtr& a,
                         const typename pcl::PointCloud<PointT>::ConstPtr& b, typename pcl::PointCloud<PointT>::Ptr& out) {
  c.initialization();
  c.setMacroblockSize(16);
  c.setColorVarThreshold(16);
  c.setMaxIterations(50);
  c.setDoICPColorOffset(false);   // bool overload: do_icp_color_offset_
  c.setDoICPColorOffset(1e-8f);   // float overload: transformationepsilon_
  std::stringstream i_data, p_data;
  c.generatePointCloudDeltaFrame(a, b, out, i_data, p_data);         // defaults: icp_on_original = false, write_out_cloud = true
  c.encodePointCloudDeltaFrame(a, b, out, i_data, p_data);           // defaults: icp_on_original = false, write_out_cloud = false
}

int main(int argc, char** argv) {
  typedef pcl::PointXYZRGB PointT;
  const int n = argc > 1 ? atoi(argv[1]) : 40000;
  const bool delta = argc > 2 && atoi(argv[2]) != 0;
  std::vector<boost::shared_ptr<pcl::PointCloud<PointT> > > group;
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / (double)(1ull << 53); };
  for (int f = 0; f < 3; ++f) {  // a sphere shell that moves a little from frame to frame (SURVEY.md cfg5 in small)
    boost::shared_ptr<pcl::PointCloud<PointT> > cloud(new pcl::PointCloud<PointT>());
    for (int i = 0; i < n; ++i) {
      const double ct = 2 * rnd() - 1, st = std::sqrt(1 - ct * ct), ph = 6.283185307179586 * rnd();
      PointT p;
      p.x = (float)(0.5 + 0.002 * f + 0.3 * st * std::cos(ph)); p.y = (float)(0.5 + 0.3 * st * std::sin(ph)); p.z = (float)(0.5 + 0.3 * ct);
      p.r = (uint8_t)(255 * p.x); p.g = (uint8_t)(255 * p.y); p.b = (uint8_t)(255 * p.z);
      cloud->push_back(p);
    }
    group.push_back(cloud);
  }
  app_like<PointT> app;
  app.complete_initialization();
  int bad = app.evaluate_group(group, delta);
  if (argc > 3) {
    pcl::PointCloud<PointT>::Ptr out(new pcl::PointCloud<PointT>());
    touch_whole_surface<PointT>(*app.encoder_V2_, group[0], group[1], out);
  }
  std::printf(bad ? "FAILED\n" : "ok\n");
  return bad;
}

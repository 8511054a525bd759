// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/oracle.py).
//
// Exact k-nearest-neighbour search over an xyz16 cloud, standing in for pcl::search::KdTree / pcl::KdTreeFLANN
// (FLANN KDTreeSingleIndex, L2_Simple<float>, eps = 0) that the upstream registration classes query
// (pcl::Registration::tree_, fast_gicp's search_source_/search_target_).  Distances are accumulated in float in
// FLANN's order ((dx*dx + dy*dy) + dz*dz); results are sorted by (distance, index) so ties are deterministic.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

namespace orc {

class KdTree {
 public:
  void build(const float* xyz16, int64_t n) {
    pts = xyz16;
    idx.resize(n);
    std::iota(idx.begin(), idx.end(), 0);
    // non-finite points never take part
    idx.erase(std::remove_if(idx.begin(), idx.end(),
                             [&](int i) { return !(std::isfinite(p(i, 0)) && std::isfinite(p(i, 1)) && std::isfinite(p(i, 2))); }),
              idx.end());
    nodes.clear();
    nodes.reserve(idx.size() / 4 + 16);
    if (!idx.empty()) build_rec(0, static_cast<int>(idx.size()));
  }

  // k nearest of query q -> (index, squared distance) ascending by (distance, index); returns the number found
  int knn(const float* q, int k, int* out_idx, float* out_d2) const {
    Heap h(k);
    if (!nodes.empty()) search(0, q, h);
    std::sort(h.items.begin(), h.items.end());
    for (size_t i = 0; i < h.items.size(); i++) {
      out_d2[i] = h.items[i].first;
      out_idx[i] = h.items[i].second;
    }
    return static_cast<int>(h.items.size());
  }

  static inline float sqdist(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;  // compiled with -ffp-contract=off: every step is rounded to float
  }

 private:
  struct Node {
    int lo, hi;        // leaf: range in idx
    int left, right;   // children (-1 for a leaf)
    int axis;
    float split;
    float bmin[3], bmax[3];
  };
  struct Heap {  // bounded max-heap on (distance, index)
    explicit Heap(int k_) : k(k_) { items.reserve(k_ + 1); }
    int k;
    std::vector<std::pair<float, int>> items;
    bool full() const { return static_cast<int>(items.size()) >= k; }
    std::pair<float, int> worst() const { return items.front(); }
    void offer(float d, int i) {
      const std::pair<float, int> c(d, i);
      if (!full()) {
        items.push_back(c);
        std::push_heap(items.begin(), items.end());
      } else if (c < items.front()) {
        std::pop_heap(items.begin(), items.end());
        items.back() = c;
        std::push_heap(items.begin(), items.end());
      }
    }
  };

  float p(int i, int a) const { return pts[static_cast<int64_t>(i) * 4 + a]; }

  int build_rec(int lo, int hi) {
    const int id = static_cast<int>(nodes.size());
    nodes.push_back(Node());
    Node nd;
    nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1; nd.axis = 0; nd.split = 0;
    for (int a = 0; a < 3; a++) { nd.bmin[a] = std::numeric_limits<float>::max(); nd.bmax[a] = -std::numeric_limits<float>::max(); }
    for (int i = lo; i < hi; i++)
      for (int a = 0; a < 3; a++) { nd.bmin[a] = std::min(nd.bmin[a], p(idx[i], a)); nd.bmax[a] = std::max(nd.bmax[a], p(idx[i], a)); }
    if (hi - lo > 10) {
      int ax = 0;
      for (int a = 1; a < 3; a++)
        if (nd.bmax[a] - nd.bmin[a] > nd.bmax[ax] - nd.bmin[ax]) ax = a;
      const int mid = (lo + hi) / 2;
      std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int x, int y) { return p(x, ax) < p(y, ax); });
      nd.axis = ax;
      nd.split = p(idx[mid], ax);
      nodes[id] = nd;
      const int l = build_rec(lo, mid);
      const int r = build_rec(mid, hi);
      nodes[id].left = l;
      nodes[id].right = r;
    } else {
      nodes[id] = nd;
    }
    return id;
  }

  // lower bound of the float squared distance from q to anything inside the node's box (same rounding sequence)
  static inline float box_sqdist(const Node& nd, const float* q) {
    float d[3];
    for (int a = 0; a < 3; a++) d[a] = std::max(std::max(nd.bmin[a] - q[a], q[a] - nd.bmax[a]), 0.0f);
    return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
  }

  void search(int id, const float* q, Heap& h) const {
    const Node& nd = nodes[id];
    if (h.full() && box_sqdist(nd, q) > h.worst().first) return;
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; i++) h.offer(sqdist(q, pts + static_cast<int64_t>(idx[i]) * 4), idx[i]);
      return;
    }
    const bool left_first = q[nd.axis] < nd.split;
    search(left_first ? nd.left : nd.right, q, h);
    search(left_first ? nd.right : nd.left, q, h);
  }

  const float* pts = nullptr;
  std::vector<int> idx;
  std::vector<Node> nodes;
};

}  // namespace orc

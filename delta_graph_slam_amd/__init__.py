"""delta_graph_slam_amd -- MI355X-native scan registration (NDT / GICP) hot path."""

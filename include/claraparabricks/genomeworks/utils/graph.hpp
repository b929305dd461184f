// graph.hpp -- small labelled graph containers returned by cudapoa::Batch::get_graphs().
// Same public surface as the reference's utils/graph.hpp (Graph, DirectedGraph, UndirectedGraph, DOT / GFA
// serialisation). Storage is ordered (std::map), so serialisation order is deterministic; the reference leaves
// it implementation-defined (unordered_map iteration).
#pragma once

#include <cstdint>
#include <map>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{

class Graph
{
public:
    using node_id_t     = int32_t;
    using edge_weight_t = int32_t;
    using edge_t        = std::pair<node_id_t, node_id_t>;

    /// Nodes reachable over one edge from `node` (empty if the node has none / does not exist).
    const std::vector<node_id_t>& get_adjacent_nodes(node_id_t node) const
    {
        static const std::vector<node_id_t> none;
        const auto it = adjacency_.find(node);
        return it == adjacency_.end() ? none : it->second;
    }

    /// Ids of all nodes that have at least one adjacent node.
    const std::vector<node_id_t> get_node_ids() const
    {
        std::vector<node_id_t> ids;
        ids.reserve(adjacency_.size());
        for (const auto& kv : adjacency_) ids.push_back(kv.first);
        return ids;
    }

    /// All edges with their weights.
    const std::vector<std::pair<edge_t, edge_weight_t>> get_edges() const { return {edges_.begin(), edges_.end()}; }

    /// First label set for a node wins (insert semantics).
    void set_node_label(node_id_t node, const std::string& label) { labels_.insert({node, label}); }

    /// Label of a node, "" if none.
    std::string get_node_label(node_id_t node) const
    {
        const auto it = labels_.find(node);
        return it == labels_.end() ? std::string() : it->second;
    }

protected:
    bool directed_edge_exists(edge_t edge) const { return edges_.count(edge) != 0; }
    void link(edge_t edge) { adjacency_[edge.first].push_back(edge.second); }

    void labels_to_dot(std::ostringstream& os) const
    {
        for (const auto& kv : labels_) os << kv.first << " [label=\"" << kv.second << "\"];\n";
    }
    void edges_to_dot(std::ostringstream& os, const char* sep) const
    {
        for (const auto& kv : edges_)
            os << kv.first.first << " " << sep << " " << kv.first.second << " [label=\"" << kv.second << "\"];\n";
    }

    std::map<node_id_t, std::vector<node_id_t>> adjacency_;
    std::map<edge_t, edge_weight_t> edges_;
    std::map<node_id_t, std::string> labels_;
};

class DirectedGraph : public Graph
{
public:
    /// Adds from -> to unless it already exists.
    void add_edge(node_id_t from, node_id_t to, edge_weight_t weight = 0)
    {
        const edge_t e(from, to);
        if (!directed_edge_exists(e))
        {
            edges_.insert({e, weight});
            link(e);
        }
    }

    std::string serialize_to_gfa() const
    {
        std::ostringstream os;
        os << "H\tVN:Z:1.0" << std::endl;
        for (const auto& kv : labels_) os << "S\t" << kv.first << "\t" << kv.second << std::endl;
        for (const auto& kv : edges_) os << "L\t" << kv.first.first << "\t+\t" << kv.first.second << "\t+\t*" << std::endl;
        return os.str();
    }

    std::string serialize_to_dot() const
    {
        std::ostringstream os;
        os << "digraph g {\n";
        labels_to_dot(os);
        edges_to_dot(os, "->");
        os << "}\n";
        return os.str();
    }
};

class UndirectedGraph : public Graph
{
public:
    void add_edge(node_id_t from, node_id_t to, edge_weight_t weight = 0)
    {
        const edge_t e(from, to), r(to, from);
        if (!directed_edge_exists(e) && !directed_edge_exists(r))
        {
            edges_.insert({e, weight});
            link(e);
            link(r);
        }
    }

    std::string serialize_to_dot() const
    {
        std::ostringstream os;
        os << "graph g {\n";
        labels_to_dot(os);
        edges_to_dot(os, "--");
        os << "}\n";
        return os.str();
    }
};

} // namespace genomeworks
} // namespace claraparabricks

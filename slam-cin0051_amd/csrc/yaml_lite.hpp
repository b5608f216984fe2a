// yaml_lite.hpp -- the subset of OpenCV FileStorage YAML-1.0 that the reference's config files
// use (test/data/*.yml): "%YAML:1.0", "---", comments, flat "Key: scalar", inline sequences
// "Key: [a, b]" and "!!opencv-matrix" maps with rows/cols/dt/data.  Replaces the cv::FileStorage
// reads at include/slam/frontend/feature_detector.hpp:54-94, src/frontend/feature_matcher.cpp:19-59,
// src/backend/loop_closure.cpp:33-81 and include/slam/common/common.hpp:78-95.
//
// cv::FileStorage semantics kept: a missing key reads as 0 / 0.0 / "" (operator>> default), an
// integer read of a real node rounds to nearest-even (cvRound), a real read of an integer node
// converts.
#pragma once

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace slamx {

struct YamlMatrix {
    int rows = 0, cols = 0;
    std::string dt;
    std::vector<double> data;
};

struct YamlNode {
    enum Kind { NONE, SCALAR, SEQ, MATRIX } kind = NONE;
    std::string scalar;  // unquoted text
    bool quoted = false;
    std::vector<double> seq;
    YamlMatrix mat;
};

class YamlFile {
public:
    bool open(const std::string& path) {
        std::ifstream in(path);
        if (!in) {
            return false;
        }
        std::vector<std::string> lines;
        std::string line;
        while (std::getline(in, line)) {
            lines.push_back(strip_comment(line));
        }
        for (size_t i = 0; i < lines.size(); ++i) {
            const std::string& l = lines[i];
            const std::string t = trim(l);
            if (t.empty() || t[0] == '%' || t == "---" || t == "...") {
                continue;
            }
            if (indent_of(l) != 0) {
                continue;  // nested content is consumed by its parent below
            }
            const size_t colon = t.find(':');
            if (colon == std::string::npos) {
                continue;
            }
            const std::string key = trim(t.substr(0, colon));
            std::string rest = trim(t.substr(colon + 1));
            YamlNode node;
            if (rest.rfind("!!opencv-matrix", 0) == 0) {
                node.kind = YamlNode::MATRIX;
                size_t j = i + 1;
                for (; j < lines.size(); ++j) {
                    const std::string tt = trim(lines[j]);
                    if (tt.empty()) {
                        continue;
                    }
                    if (indent_of(lines[j]) == 0) {
                        break;
                    }
                    const size_t c2 = tt.find(':');
                    if (c2 == std::string::npos) {
                        continue;
                    }
                    const std::string k2 = trim(tt.substr(0, c2));
                    std::string v2 = trim(tt.substr(c2 + 1));
                    if (k2 == "rows") {
                        node.mat.rows = std::atoi(v2.c_str());
                    } else if (k2 == "cols") {
                        node.mat.cols = std::atoi(v2.c_str());
                    } else if (k2 == "dt") {
                        node.mat.dt = unquote(v2);
                    } else if (k2 == "data") {
                        // the sequence may continue over following lines until ']'
                        std::string all = v2;
                        size_t m = j;
                        while (all.find(']') == std::string::npos && m + 1 < lines.size()) {
                            ++m;
                            all += " " + trim(lines[m]);
                        }
                        j = m;
                        node.mat.data = parse_seq(all);
                    }
                }
                i = j - 1;
            } else if (!rest.empty() && rest[0] == '[') {
                std::string all = rest;
                size_t m = i;
                while (all.find(']') == std::string::npos && m + 1 < lines.size()) {
                    ++m;
                    all += " " + trim(lines[m]);
                }
                i = m;
                node.kind = YamlNode::SEQ;
                node.seq = parse_seq(all);
            } else {
                node.kind = YamlNode::SCALAR;
                node.quoted = !rest.empty() && (rest[0] == '"' || rest[0] == '\'');
                node.scalar = unquote(rest);
            }
            nodes_[key] = node;
        }
        return true;
    }

    bool has(const std::string& key) const { return nodes_.count(key) != 0; }

    int get_int(const std::string& key) const {
        auto it = nodes_.find(key);
        if (it == nodes_.end() || it->second.kind != YamlNode::SCALAR) {
            return 0;
        }
        const std::string& s = it->second.scalar;
        if (s.find_first_of(".eE") != std::string::npos && s.find_first_of("xX") == std::string::npos) {
            return static_cast<int>(std::lrint(std::strtod(s.c_str(), nullptr)));  // cvRound
        }
        return static_cast<int>(std::strtol(s.c_str(), nullptr, 0));
    }
    double get_double(const std::string& key) const {
        auto it = nodes_.find(key);
        if (it == nodes_.end() || it->second.kind != YamlNode::SCALAR) {
            return 0.0;
        }
        return std::strtod(it->second.scalar.c_str(), nullptr);
    }
    std::string get_string(const std::string& key) const {
        auto it = nodes_.find(key);
        if (it == nodes_.end() || it->second.kind != YamlNode::SCALAR) {
            return std::string();
        }
        return it->second.scalar;
    }
    const YamlNode* node(const std::string& key) const {
        auto it = nodes_.find(key);
        return it == nodes_.end() ? nullptr : &it->second;
    }

private:
    static std::string strip_comment(const std::string& l) {
        bool in_q = false;
        char q = 0;
        for (size_t i = 0; i < l.size(); ++i) {
            const char c = l[i];
            if (in_q) {
                if (c == q) {
                    in_q = false;
                }
            } else if (c == '"' || c == '\'') {
                in_q = true;
                q = c;
            } else if (c == '#') {
                return l.substr(0, i);
            }
        }
        return l;
    }
    static std::string trim(const std::string& s) {
        const size_t a = s.find_first_not_of(" \t\r\n");
        if (a == std::string::npos) {
            return std::string();
        }
        const size_t b = s.find_last_not_of(" \t\r\n");
        return s.substr(a, b - a + 1);
    }
    static size_t indent_of(const std::string& l) {
        const size_t a = l.find_first_not_of(" \t");
        return a == std::string::npos ? 0 : a;
    }
    static std::string unquote(const std::string& s) {
        if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) {
            return s.substr(1, s.size() - 2);
        }
        return s;
    }
    static std::vector<double> parse_seq(const std::string& text) {
        std::vector<double> out;
        const size_t a = text.find('[');
        const size_t b = text.find(']');
        if (a == std::string::npos) {
            return out;
        }
        std::string body = text.substr(a + 1, (b == std::string::npos ? text.size() : b) - a - 1);
        for (auto& c : body) {
            if (c == ',') {
                c = ' ';
            }
        }
        std::istringstream ss(body);
        std::string tok;
        while (ss >> tok) {
            out.push_back(std::strtod(tok.c_str(), nullptr));
        }
        return out;
    }

    std::map<std::string, YamlNode> nodes_;
};

}  // namespace slamx

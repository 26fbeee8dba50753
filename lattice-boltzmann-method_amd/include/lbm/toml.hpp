// Minimal TOML-subset reader for the engine's host side.  The reference parses its inputs
// with toml++ (an external, un-vendored dependency absent from this image); the engine needs
// only what the reference's files use: [tables] (bare or quoted names), key = value with
// floats / integers / strings / booleans, flat numeric arrays (possibly multi-line), comments.
// The access pattern mirrors the reference's call sites, e.g. src/params.cpp:11
//   tbl["flow"]["initial_density"].value<double>()   ->   std::optional<double>
// and src/ibm.cpp:78-79  tbl[name]["x"].as_array().
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <variant>
#include <vector>

namespace lbm::toml {

struct parse_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct value_t {
  std::variant<std::monostate, double, long long, std::string, bool, std::vector<double>> v;
  bool is_int = false;
};

class array {
 public:
  explicit array(const std::vector<double>* d) : d_(d) {}
  size_t size() const { return d_->size(); }
  struct elem {
    double x;
    template <class T>
    std::optional<T> value() const { return static_cast<T>(x); }
  };
  elem at(size_t i) const { return elem{d_->at(i)}; }
  const std::vector<double>& values() const { return *d_; }

 private:
  const std::vector<double>* d_;
};

class table;

// node_view: result of operator[]; empty when the key does not exist
class node_view {
 public:
  node_view() = default;
  explicit node_view(const value_t* v) : val_(v) {}
  explicit node_view(const table* t) : tbl_(t) {}
  node_view operator[](const std::string& key) const;
  explicit operator bool() const { return val_ || tbl_; }

  template <class T>
  std::optional<T> value() const {
    if (!val_) return std::nullopt;
    if constexpr (std::is_same_v<T, std::string>) {
      if (auto p = std::get_if<std::string>(&val_->v)) return *p;
      return std::nullopt;
    } else if constexpr (std::is_same_v<T, bool>) {
      if (auto p = std::get_if<bool>(&val_->v)) return *p;
      return std::nullopt;
    } else {
      if (auto p = std::get_if<double>(&val_->v)) return static_cast<T>(*p);
      if (auto p = std::get_if<long long>(&val_->v)) return static_cast<T>(*p);
      return std::nullopt;
    }
  }
  std::optional<array> as_array() const {
    if (!val_) return std::nullopt;
    if (auto p = std::get_if<std::vector<double>>(&val_->v)) return array(p);
    return std::nullopt;
  }

 private:
  const value_t* val_ = nullptr;
  const table* tbl_ = nullptr;
};

class table {
 public:
  node_view operator[](const std::string& key) const {
    if (auto it = subs_.find(key); it != subs_.end()) return node_view(&it->second);
    if (auto it = vals_.find(key); it != vals_.end()) return node_view(&it->second);
    return node_view();
  }
  bool contains(const std::string& key) const { return subs_.count(key) || vals_.count(key); }
  std::map<std::string, value_t> vals_;
  std::map<std::string, table> subs_;
};

inline node_view node_view::operator[](const std::string& key) const {
  return tbl_ ? (*tbl_)[key] : node_view();
}

namespace detail {
inline std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}
inline std::string strip_comment(const std::string& s) {
  bool in_str = false;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '"') in_str = !in_str;
    if (s[i] == '#' && !in_str) return s.substr(0, i);
  }
  return s;
}
inline std::string unquote(const std::string& s) {
  if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front())
    return s.substr(1, s.size() - 2);
  return s;
}
inline double parse_number(std::string t, bool* is_int, int line) {
  std::string clean;
  for (char ch : t)
    if (ch != '_') clean += ch;
  char* end = nullptr;
  const double d = std::strtod(clean.c_str(), &end);
  if (end == clean.c_str() || *end != '\0')
    throw parse_error("line " + std::to_string(line) + ": cannot parse value '" + t + "'");
  if (is_int) *is_int = clean.find_first_of(".eEni") == std::string::npos;
  return d;
}
}  // namespace detail

inline table parse(std::istream& in) {
  using namespace detail;
  table root;
  table* cur = &root;
  std::string raw;
  int line = 0;
  while (std::getline(in, raw)) {
    ++line;
    std::string s = trim(strip_comment(raw));
    if (s.empty()) continue;
    if (s.front() == '[') {
      if (s.back() != ']') throw parse_error("line " + std::to_string(line) + ": unterminated table header");
      cur = &root.subs_[unquote(trim(s.substr(1, s.size() - 2)))];
      continue;
    }
    const size_t eq = s.find('=');
    if (eq == std::string::npos) throw parse_error("line " + std::to_string(line) + ": expected key = value");
    const std::string key = unquote(trim(s.substr(0, eq)));
    std::string val = trim(s.substr(eq + 1));
    value_t out;
    if (!val.empty() && val.front() == '[') {  // array, may continue over lines
      while (val.find(']') == std::string::npos) {
        if (!std::getline(in, raw)) throw parse_error("line " + std::to_string(line) + ": unterminated array");
        ++line;
        val += " " + trim(strip_comment(raw));
      }
      std::vector<double> a;
      std::string body = val.substr(1, val.rfind(']') - 1), tok;
      std::stringstream ss(body);
      while (std::getline(ss, tok, ',')) {
        tok = trim(tok);
        if (!tok.empty()) a.push_back(parse_number(tok, nullptr, line));
      }
      out.v = std::move(a);
    } else if (!val.empty() && (val.front() == '"' || val.front() == '\'')) {
      out.v = unquote(val);
    } else if (val == "true" || val == "false") {
      out.v = (val == "true");
    } else {
      bool is_int = false;
      const double d = parse_number(val, &is_int, line);
      if (is_int) out.v = (long long)d;
      else out.v = d;
      out.is_int = is_int;
    }
    cur->vals_[key] = std::move(out);
  }
  return root;
}

inline table parse_file(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw parse_error("cannot open " + path);
  return parse(f);
}

}  // namespace lbm::toml

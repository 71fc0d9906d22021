// toml_lite.h — the TOML subset the reference's config files use (simulation.toml, template.toml): comments, [tables],
// [[arrays of tables]], bare / quoted / dotted keys, strings, integers, floats (inf / nan), booleans, (multi-line)
// arrays and inline tables.  No dates, no multi-line strings.  Parse errors throw std::runtime_error with a line number,
// which the CLI reports the way `toml::from_str` errors surface through `?` in cli.rs:12.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace toml_lite {

struct Value;
using Table = std::map<std::string, std::shared_ptr<Value>>;
using Array = std::vector<std::shared_ptr<Value>>;

struct Value {
    enum Kind { STRING, INT, FLOAT, BOOL, ARRAY, TABLE } kind = TABLE;
    std::string s;
    int64_t i = 0;
    double f = 0;
    bool b = false;
    Array arr;
    Table tab;
    bool is_table_array = false;  // created by [[...]]
    bool is_number() const { return kind == INT || kind == FLOAT; }
    double number() const { return kind == INT ? (double)i : f; }
};

class Parser {
public:
    explicit Parser(const std::string& text) : t_(text) {}

    std::shared_ptr<Value> parse()
    {
        auto root = std::make_shared<Value>();
        Value* cur = root.get();
        for (;;) {
            skip_ws_comments_newlines();
            if (eof()) break;
            if (peek() == '[') {
                ++p_;
                bool arr = false;
                if (peek() == '[') {
                    arr = true;
                    ++p_;
                }
                skip_inline_ws();
                std::vector<std::string> path = parse_key_path();
                skip_inline_ws();
                expect(']');
                if (arr) expect(']');
                cur = arr ? open_table_array(root.get(), path) : open_table(root.get(), path);
                end_of_line();
            } else {
                std::vector<std::string> path = parse_key_path();
                skip_inline_ws();
                expect('=');
                skip_inline_ws();
                auto v = parse_value();
                assign(cur, path, v);
                end_of_line();
            }
        }
        return root;
    }

private:
    const std::string& t_;
    size_t p_ = 0;
    int line_ = 1;

    [[noreturn]] void fail(const std::string& msg) const { throw std::runtime_error("TOML parse error at line " + std::to_string(line_) + ": " + msg); }
    bool eof() const { return p_ >= t_.size(); }
    char peek() const { return eof() ? '\0' : t_[p_]; }
    void expect(char c)
    {
        if (peek() != c) fail(std::string("expected '") + c + "'");
        ++p_;
    }
    void skip_inline_ws()
    {
        while (!eof() && (t_[p_] == ' ' || t_[p_] == '\t')) ++p_;
    }
    void skip_comment()
    {
        if (peek() == '#')
            while (!eof() && t_[p_] != '\n') ++p_;
    }
    void skip_ws_comments_newlines()
    {
        for (;;) {
            skip_inline_ws();
            skip_comment();
            if (peek() == '\r') ++p_;
            if (peek() == '\n') {
                ++p_;
                ++line_;
                continue;
            }
            break;
        }
    }
    void end_of_line()
    {
        skip_inline_ws();
        skip_comment();
        if (peek() == '\r') ++p_;
        if (eof()) return;
        if (peek() != '\n') fail("unexpected characters after value");
        ++p_;
        ++line_;
    }
    static bool bare(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'; }

    std::string parse_basic_string()
    {
        char q = t_[p_++];
        std::string out;
        while (!eof() && t_[p_] != q) {
            char c = t_[p_++];
            if (c == '\n') fail("newline in string");
            if (q == '"' && c == '\\') {
                if (eof()) fail("bad escape");
                char e = t_[p_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    default: fail("unsupported escape");
                }
            } else {
                out += c;
            }
        }
        if (eof()) fail("unterminated string");
        ++p_;
        return out;
    }
    std::vector<std::string> parse_key_path()
    {
        std::vector<std::string> path;
        for (;;) {
            skip_inline_ws();
            if (peek() == '"' || peek() == '\'') {
                path.push_back(parse_basic_string());
            } else {
                size_t b = p_;
                while (!eof() && bare(t_[p_])) ++p_;
                if (p_ == b) fail("expected a key");
                path.push_back(t_.substr(b, p_ - b));
            }
            skip_inline_ws();
            if (peek() == '.') {
                ++p_;
                continue;
            }
            break;
        }
        return path;
    }
    Value* child_table(Value* t, const std::string& k)
    {
        auto it = t->tab.find(k);
        if (it == t->tab.end()) {
            auto v = std::make_shared<Value>();
            v->kind = Value::TABLE;
            t->tab[k] = v;
            return v.get();
        }
        Value* v = it->second.get();
        if (v->kind == Value::ARRAY && v->is_table_array) {
            if (v->arr.empty()) fail("empty table array");
            return v->arr.back().get();
        }
        if (v->kind != Value::TABLE) fail("key `" + k + "` is not a table");
        return v;
    }
    Value* open_table(Value* root, const std::vector<std::string>& path)
    {
        Value* t = root;
        for (auto& k : path) t = child_table(t, k);
        return t;
    }
    Value* open_table_array(Value* root, const std::vector<std::string>& path)
    {
        Value* t = root;
        for (size_t i = 0; i + 1 < path.size(); ++i) t = child_table(t, path[i]);
        auto& slot = t->tab[path.back()];
        if (!slot) {
            slot = std::make_shared<Value>();
            slot->kind = Value::ARRAY;
            slot->is_table_array = true;
        }
        if (slot->kind != Value::ARRAY || !slot->is_table_array) fail("key `" + path.back() + "` is not an array of tables");
        auto el = std::make_shared<Value>();
        el->kind = Value::TABLE;
        slot->arr.push_back(el);
        return el.get();
    }
    void assign(Value* t, const std::vector<std::string>& path, const std::shared_ptr<Value>& v)
    {
        for (size_t i = 0; i + 1 < path.size(); ++i) t = child_table(t, path[i]);
        if (t->tab.count(path.back())) fail("duplicate key `" + path.back() + "`");
        t->tab[path.back()] = v;
    }
    std::shared_ptr<Value> parse_value()
    {
        auto v = std::make_shared<Value>();
        char c = peek();
        if (c == '"' || c == '\'') {
            v->kind = Value::STRING;
            v->s = parse_basic_string();
        } else if (c == '[') {
            ++p_;
            v->kind = Value::ARRAY;
            for (;;) {
                skip_ws_comments_newlines();
                if (peek() == ']') {
                    ++p_;
                    break;
                }
                v->arr.push_back(parse_value());
                skip_ws_comments_newlines();
                if (peek() == ',') {
                    ++p_;
                    continue;
                }
                skip_ws_comments_newlines();
                if (peek() == ']') {
                    ++p_;
                    break;
                }
                fail("expected ',' or ']' in array");
            }
        } else if (c == '{') {
            ++p_;
            v->kind = Value::TABLE;
            skip_inline_ws();
            if (peek() == '}') {
                ++p_;
                return v;
            }
            for (;;) {
                skip_inline_ws();
                std::vector<std::string> path = parse_key_path();
                skip_inline_ws();
                expect('=');
                skip_inline_ws();
                assign(v.get(), path, parse_value());
                skip_inline_ws();
                if (peek() == ',') {
                    ++p_;
                    continue;
                }
                if (peek() == '}') {
                    ++p_;
                    break;
                }
                fail("expected ',' or '}' in inline table");
            }
        } else {
            size_t b = p_;
            while (!eof() && t_[p_] != ',' && t_[p_] != ']' && t_[p_] != '}' && t_[p_] != '#' && t_[p_] != '\n' && t_[p_] != '\r' &&
                   t_[p_] != ' ' && t_[p_] != '\t')
                ++p_;
            std::string tok = t_.substr(b, p_ - b);
            if (tok == "true" || tok == "false") {
                v->kind = Value::BOOL;
                v->b = tok == "true";
                return v;
            }
            std::string clean;
            for (char ch : tok)
                if (ch != '_') clean += ch;
            if (clean.empty()) fail("expected a value");
            std::string body = (clean[0] == '+' || clean[0] == '-') ? clean.substr(1) : clean;
            if (body == "inf" || body == "nan") {
                v->kind = Value::FLOAT;
                v->f = body == "inf" ? INFINITY : NAN;
                if (clean[0] == '-') v->f = -v->f;
                return v;
            }
            bool is_float = clean.find_first_of(".eE") != std::string::npos;
            char* end = nullptr;
            if (is_float) {
                v->kind = Value::FLOAT;
                v->f = std::strtod(clean.c_str(), &end);
            } else {
                v->kind = Value::INT;
                v->i = std::strtoll(clean.c_str(), &end, 10);
            }
            if (!end || *end != '\0') fail("invalid value `" + tok + "`");
        }
        return v;
    }
};

inline std::shared_ptr<Value> parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace toml_lite

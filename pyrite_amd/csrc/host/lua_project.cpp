// lua_project.cpp -- reads Pyrite project files (*.lua) into the typed tree of pyrite_host.hpp.
//
// The reference runs a project file through mlua with the prelude pyrite/src/project/lib.lua, which turns the script into
// plain tables tagged with `type` (project/mod.rs:55-100), and `typed_nodes::FromLua` turns those into the typed tree
// (project/mod.rs:103-252). There is no Lua here, so this file does the same in two steps:
//   1. an evaluator for the declarative subset of Lua that project files are written in -- local and global assignments,
//      `return`, table constructors, calls with parenthesised / table / string arguments, method calls (`:with{}`, `:clone()`,
//      `:mix()`), field chains, arithmetic, `..`, comparisons, `and` / `or` / `not`, `#`, `require` -- with the prelude's
//      functions built in (they produce the same tagged tables as lib.lua does);
//   2. the FromLua step: tagged tables -> pyrite::Project, one Expression node per Lua table (identity is what gives one
//      spectrum / texture id per table, project/spectra.rs:116-145).
// Control flow and function definitions -- which no project file under pyrite/test uses -- are rejected with the file and
// line. The Python front-end (pyrite_amd/lua_project.py) does the same job; tests/test_host_cpp.py holds the two together.
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <unordered_map>

#include "pyrite_host.hpp"

namespace pyrite {
namespace {

struct LuaError : ProjectError {
    using ProjectError::ProjectError;
};

// ------------------------------------------------------------------------------------------------ values
struct Table;
using TablePtr = std::shared_ptr<Table>;
struct Value {
    enum Kind { Nil, Bool, Number, String, Tab, Function, Method } kind = Nil;
    bool b = false;
    double n = 0.0;
    std::string s; // String: the text; Function: the builtin's name; Method: the method's name
    TablePtr t;    // Tab; Method: the receiver
    static Value number(double x) {
        Value v;
        v.kind = Number, v.n = x;
        return v;
    }
    static Value string(std::string x) {
        Value v;
        v.kind = String, v.s = std::move(x);
        return v;
    }
    static Value boolean(bool x) {
        Value v;
        v.kind = Bool, v.b = x;
        return v;
    }
    static Value table(TablePtr x) {
        Value v;
        v.kind = Tab, v.t = std::move(x);
        return v;
    }
    static Value function(std::string name) {
        Value v;
        v.kind = Function, v.s = std::move(name);
        return v;
    }
    bool truthy() const { return !(kind == Nil || (kind == Bool && !b)); }
    const char* type_name() const {
        switch (kind) {
        case Nil: return "nil";
        case Bool: return "boolean";
        case Number: return "number";
        case String: return "string";
        case Tab: return "table";
        default: return "function";
        }
    }
};
struct Table {
    std::vector<Value> items;                           // positional part
    std::vector<std::pair<std::string, Value>> fields;  // named part, in insertion order
    const Value* find(const std::string& key) const {
        for (auto& kv : fields)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    Value get(const std::string& key) const {
        const Value* v = find(key);
        return v ? *v : Value();
    }
    void set(const std::string& key, Value v) {
        for (auto& kv : fields)
            if (kv.first == key) {
                kv.second = std::move(v);
                return;
            }
        fields.emplace_back(key, std::move(v));
    }
    std::string type() const {
        const Value* v = find("type");
        return v && v->kind == Value::String ? v->s : std::string();
    }
};

// ------------------------------------------------------------------------------------------------ tokens
struct Token {
    enum Kind { Number, Name, Keyword, String, Op, Eof } kind;
    std::string text;
    double number = 0.0;
    int line = 1;
};
const std::set<std::string> kKeywords = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "goto", "if", "in",
                                         "local", "nil", "not", "or", "repeat", "return", "then", "true", "until", "while"};

std::vector<Token> tokenize(const std::string& text, const std::string& name) {
    std::vector<Token> out;
    size_t pos = 0;
    int line = 1;
    auto fail = [&](const std::string& what) { throw LuaError(name + ":" + std::to_string(line) + ": " + what); };
    auto long_bracket = [&](size_t at, size_t& level) -> bool { // is text[at] the start of [[ or [=[ ... ?
        if (text[at] != '[') return false;
        size_t k = at + 1;
        while (k < text.size() && text[k] == '=') ++k;
        if (k < text.size() && text[k] == '[') {
            level = k - at - 1;
            return true;
        }
        return false;
    };
    auto read_long = [&](size_t at, size_t level) -> std::string { // at = first char after the opening bracket; moves pos past the close
        const std::string close = "]" + std::string(level, '=') + "]";
        const size_t end = text.find(close, at);
        if (end == std::string::npos) fail("unfinished long string / comment");
        std::string body = text.substr(at, end - at);
        pos = end + close.size();
        return body;
    };
    while (pos < text.size()) {
        const char c = text[pos];
        if (c == '\n') {
            ++line, ++pos;
            continue;
        }
        if (std::isspace((unsigned char)c)) {
            ++pos;
            continue;
        }
        if (c == '-' && pos + 1 < text.size() && text[pos + 1] == '-') {
            size_t level = 0;
            if (pos + 2 < text.size() && long_bracket(pos + 2, level)) {
                const std::string body = read_long(pos + 2 + level + 2, level);
                for (char ch : body) line += ch == '\n';
            } else {
                while (pos < text.size() && text[pos] != '\n') ++pos;
            }
            continue;
        }
        Token tok;
        tok.line = line;
        if (std::isdigit((unsigned char)c) || (c == '.' && pos + 1 < text.size() && std::isdigit((unsigned char)text[pos + 1]))) {
            char* end = nullptr;
            if (c == '0' && pos + 1 < text.size() && (text[pos + 1] == 'x' || text[pos + 1] == 'X'))
                tok.number = (double)std::strtoull(text.c_str() + pos, &end, 16);
            else
                tok.number = std::strtod(text.c_str() + pos, &end);
            tok.kind = Token::Number;
            pos = (size_t)(end - text.c_str());
        } else if (std::isalpha((unsigned char)c) || c == '_') {
            size_t e = pos;
            while (e < text.size() && (std::isalnum((unsigned char)text[e]) || text[e] == '_')) ++e;
            tok.text = text.substr(pos, e - pos);
            tok.kind = kKeywords.count(tok.text) ? Token::Keyword : Token::Name;
            pos = e;
        } else if (c == '"' || c == '\'') {
            std::string body;
            size_t e = pos + 1;
            for (;; ++e) {
                if (e >= text.size() || text[e] == '\n') fail("unfinished string");
                if (text[e] == c) break;
                if (text[e] == '\\' && e + 1 < text.size()) {
                    const char x = text[++e];
                    switch (x) {
                    case 'n': body += '\n'; break;
                    case 't': body += '\t'; break;
                    case 'r': body += '\r'; break;
                    case 'a': body += '\a'; break;
                    case 'b': body += '\b'; break;
                    case 'f': body += '\f'; break;
                    case 'v': body += '\v'; break;
                    case '0': body += '\0'; break;
                    default: body += x;
                    }
                } else {
                    body += text[e];
                }
            }
            tok.kind = Token::String;
            tok.text = body;
            pos = e + 1;
        } else {
            size_t level = 0;
            if (c == '[' && long_bracket(pos, level)) {
                std::string body = read_long(pos + level + 2, level);
                for (char ch : body) line += ch == '\n';
                if (!body.empty() && body[0] == '\n') body.erase(0, 1);
                tok.kind = Token::String;
                tok.text = body;
            } else {
                static const char* ops[] = {"...", "..", "==", "~=", "<=", ">="};
                tok.kind = Token::Op;
                for (const char* op : ops)
                    if (text.compare(pos, std::strlen(op), op) == 0) {
                        tok.text = op;
                        break;
                    }
                if (tok.text.empty()) {
                    if (std::strchr("-+*/%^#<>=(){}[];:,.", c) == nullptr) fail(std::string("unexpected character '") + c + "'");
                    tok.text = std::string(1, c);
                }
                pos += tok.text.size();
            }
        }
        out.push_back(tok);
    }
    Token eof;
    eof.kind = Token::Eof;
    eof.line = line;
    out.push_back(eof);
    return out;
}

// ------------------------------------------------------------------------------------------------ the prelude (lib.lua)
// A builtin takes its arguments either as one table with named fields (`shape.sphere {radius = 1}`) or positionally
// (`vector(0, 1, 0)`); the result is a table tagged with `type` that has exactly the fields the typed tree knows -- fields it
// does not know are dropped, as typed_nodes ignores them (dragon.lua's `_ior`, the scenes' `spectrum_bins`).
struct Builtin {
    const char* name;
    const char* type;
    std::vector<const char*> params;
    bool keep_extra; // ray-marched shapes keep whatever they are given (they are rejected later)
};
const std::vector<Builtin>& builtins() {
    static const std::vector<Builtin> list = {
        {"vector", "vector", {"x", "y", "z", "w"}, false},
        {"rgb", "rgb", {"red", "green", "blue"}, false},
        {"spectrum", "spectrum", {"format", "min", "max", "points", "name"}, false},
        {"blackbody", "blackbody", {"temperature"}, false},
        {"fresnel", "fresnel", {"ior", "env_ior"}, false},
        {"mix", "mix", {"lhs", "rhs", "amount"}, false},
        {"material.diffuse", "diffuse", {"color"}, false},
        {"material.emissive", "emissive", {"color"}, false},
        {"material.mirror", "mirror", {"color"}, false},
        {"material.refractive", "refractive", {"color", "ior", "dispersion", "env_ior", "env_dispersion"}, false},
        {"shape.sphere", "sphere", {"position", "radius", "material", "texture_scale"}, false},
        {"shape.plane", "plane", {"origin", "normal", "material", "texture_scale"}, false},
        {"shape.mesh", "mesh", {"file", "materials", "scale", "transform"}, false},
        {"shape.ray_marched", "ray_marched", {}, true},
        {"ray_marched.quaternion_julia", "quaternion_julia", {}, true},
        {"ray_marched.mandelbulb", "mandelbulb", {}, true},
        {"bounds.box", "box", {"min", "max"}, false},
        {"light.point", "point_light", {"position", "color"}, false},
        {"light.directional", "directional_light", {"direction", "width", "color"}, false},
        {"transform.look_at", "look_at", {"from", "to", "up"}, false},
        {"camera.perspective", "perspective", {"transform", "fov", "focus_distance", "aperture"}, false},
        {"renderer.simple", "simple", {"pixel_samples", "threads", "bounces", "light_samples", "spectrum_samples", "spectrum_resolution", "tile_size"}, false},
        {"renderer.bidirectional", "bidirectional", {"pixel_samples", "threads", "bounces", "light_samples", "spectrum_samples", "spectrum_resolution", "tile_size"}, false},
        {"renderer.photon_mapping", "photon_mapping", {"pixel_samples", "threads", "bounces", "light_samples", "spectrum_samples", "spectrum_resolution", "tile_size"}, false},
    };
    return list;
}

Value call_builtin(const std::string& name, const std::vector<Value>& args) {
    if (name == "texture") { // lib.lua:178-195: texture(path, "linear", "mono")
        if (args.empty() || args[0].kind != Value::String) throw LuaError("texture expects a path");
        auto t = std::make_shared<Table>();
        bool linear = false, mono = false;
        for (size_t i = 1; i < args.size(); ++i) {
            linear = linear || (args[i].kind == Value::String && args[i].s == "linear");
            mono = mono || (args[i].kind == Value::String && args[i].s == "mono");
        }
        t->set("type", Value::string(mono ? "mono_texture" : "color_texture"));
        t->set("path", args[0]);
        t->set("linear", Value::boolean(linear));
        t->set("mono", Value::boolean(mono));
        return Value::table(t);
    }
    for (const Builtin& b : builtins()) {
        if (name != b.name) continue;
        auto t = std::make_shared<Table>();
        t->set("type", Value::string(b.type));
        const bool named = args.size() == 1 && args[0].kind == Value::Tab && args[0].t->items.empty() && args[0].t->type().empty();
        if (named) {
            if (b.keep_extra)
                for (auto& kv : args[0].t->fields) t->set(kv.first, kv.second);
            for (const char* p : b.params) t->set(p, args[0].t->get(p));
        } else {
            for (size_t i = 0; i < b.params.size(); ++i) t->set(b.params[i], i < args.size() ? args[i] : Value());
        }
        // defaults the prelude fills in (lib.lua:120-176)
        if (name == "vector")
            for (const char* p : {"x", "y", "z", "w"})
                if (t->get(p).kind == Value::Nil) t->set(p, Value::number(0.0));
        if (name == "rgb")
            for (const char* p : {"red", "green", "blue"})
                if (t->get(p).kind == Value::Nil) t->set(p, Value::number(0.0));
        if (name == "fresnel" && t->get("env_ior").kind == Value::Nil) t->set("env_ior", Value::number(1.0));
        if (name == "spectrum" && t->get("format").kind == Value::Nil) t->set("format", Value::string("array"));
        return Value::table(t);
    }
    throw LuaError("attempt to call a nil value");
}

struct Environment {
    std::map<std::string, Value> globals;
    std::map<std::string, Value> modules; // by normalised path
    Environment() {
        for (const char* f : {"vector", "rgb", "spectrum", "blackbody", "fresnel", "mix", "texture", "require"}) globals[f] = Value::function(f);
        auto ns = [&](const char* name, std::vector<const char*> members) {
            auto t = std::make_shared<Table>();
            for (const char* m : members) t->set(m, Value::function(std::string(name) + "." + m));
            globals[name] = Value::table(t);
        };
        ns("material", {"diffuse", "emissive", "mirror", "refractive"});
        ns("shape", {"sphere", "plane", "mesh", "ray_marched"});
        ns("ray_marched", {"quaternion_julia", "mandelbulb"});
        ns("bounds", {"box"});
        ns("light", {"point", "directional"});
        ns("transform", {"look_at"});
        ns("camera", {"perspective"});
        ns("renderer", {"simple", "bidirectional", "photon_mapping"});
        auto ls = std::make_shared<Table>(); // lib.lua:254-258: one table each, so one spectrum id each
        for (const char* n : {"d65", "a"}) {
            auto s = std::make_shared<Table>();
            s->set("type", Value::string("spectrum"));
            s->set("name", Value::string(n));
            ls->set(n, Value::table(s));
        }
        globals["light_source"] = Value::table(ls);
        auto qj = std::make_shared<Table>();
        auto cubic = std::make_shared<Table>();
        cubic->set("type", Value::string("quaternion_julia"));
        cubic->set("name", Value::string("cubic"));
        qj->set("cubic", Value::table(cubic));
        globals["quaternion_julia"] = Value::table(qj);
    }
};

// ------------------------------------------------------------------------------------------------ parser / evaluator
class Evaluator {
  public:
    Evaluator(const std::string& text, std::string name, Environment& env, std::string base_dir, std::set<std::string> loading)
        : tokens_(tokenize(text, name)), name_(std::move(name)), env_(env), base_dir_(std::move(base_dir)), loading_(std::move(loading)) {}

    Value run() {
        for (;;) {
            const Token& tok = peek();
            if (tok.kind == Token::Eof) return Value();
            if (accept_op(";")) continue;
            if (tok.kind == Token::Keyword) {
                if (tok.text == "return") {
                    next();
                    Value value;
                    if (peek().kind != Token::Eof) value = expression_list()[0]; // a chunk's first return value
                    accept_op(";");
                    if (peek().kind != Token::Eof) fail("'return' must be the last statement");
                    return value;
                }
                if (tok.text == "local") {
                    next();
                    if (peek().kind == Token::Keyword && peek().text == "function") fail("function definitions are not supported in project files");
                    std::vector<std::string> names{expect(Token::Name).text};
                    while (accept_op(",")) names.push_back(expect(Token::Name).text);
                    std::vector<Value> values;
                    if (accept_op("=")) values = expression_list();
                    for (size_t i = 0; i < names.size(); ++i) locals_[names[i]] = i < values.size() ? values[i] : Value();
                    continue;
                }
                fail("'" + tok.text + "' is not supported in project files (only assignments, calls and return are)");
            }
            Target target;
            suffixed(&target);
            if (target.valid && peek().kind == Token::Op && peek().text == "=") {
                next();
                Value value = expression();
                if (!target.container) {
                    (locals_.count(target.key) ? locals_ : env_.globals)[target.key] = value;
                } else {
                    target.container->set(target.key, value);
                }
            }
        }
    }

  private:
    struct Target {
        bool valid = false;
        TablePtr container; // null: a variable
        std::string key;
    };
    std::vector<Token> tokens_;
    size_t pos_ = 0;
    std::string name_;
    Environment& env_;
    std::map<std::string, Value> locals_;
    std::string base_dir_;
    std::set<std::string> loading_;

    const Token& peek() const { return tokens_[pos_]; }
    const Token& next() { return tokens_[pos_++]; }
    int depth_ = 0; // nesting of expression() calls (see there)
    [[noreturn]] void fail(const std::string& message) const { throw LuaError(name_ + ":" + std::to_string(peek().line) + ": " + message); }
    bool accept_op(const char* op) {
        if (peek().kind == Token::Op && peek().text == op) {
            ++pos_;
            return true;
        }
        return false;
    }
    void expect_op(const char* op) {
        if (!accept_op(op)) fail(std::string("expected ") + op + ", found '" + (peek().kind == Token::Eof ? "end of file" : peek().text) + "'");
    }
    const Token& expect(Token::Kind kind) {
        if (peek().kind != kind) fail("unexpected '" + (peek().kind == Token::Eof ? std::string("end of file") : peek().text) + "'");
        return next();
    }

    std::vector<Value> expression_list() {
        std::vector<Value> values{expression()};
        while (accept_op(",")) values.push_back(expression());
        return values;
    }

    // binary operators, loosest first -- Lua 5.3 reference manual 3.4.8
    Value expression(int level = 0) {
        static const std::vector<std::pair<std::vector<std::string>, bool>> levels = {
            {{"or"}, false}, {{"and"}, false}, {{"<", ">", "<=", ">=", "~=", "=="}, false}, {{".."}, true}, {{"+", "-"}, false}, {{"*", "/", "%"}, false}};
        if (level == (int)levels.size()) return unary();
        // Lua itself refuses chunks nested deeper than 200 levels ("chunk has too many syntax levels", LUAI_MAXCCALLS); without a
        // bound a file of 5000 opening braces ran this recursive descent off the stack (found by tests/host_asan_driver.cpp)
        struct Depth {
            int& d;
            explicit Depth(int& d_) : d(d_) { ++d; }
            ~Depth() { --d; }
        } guard(depth_);
        if (level == 0 && depth_ > 200 * ((int)levels.size() + 1)) fail("chunk has too many syntax levels");
        Value lhs = expression(level + 1);
        for (;;) {
            const Token& tok = peek();
            bool match = false;
            if (tok.kind == Token::Op || tok.kind == Token::Keyword)
                for (auto& op : levels[level].first) match = match || tok.text == op;
            if (!match) return lhs;
            const std::string op = next().text;
            Value rhs = expression(levels[level].second ? level : level + 1);
            lhs = binary(op, lhs, rhs);
            if (levels[level].second) return lhs;
        }
    }
    Value unary() {
        const Token& tok = peek();
        if (tok.kind == Token::Op && tok.text == "-") {
            next();
            Value v = unary();
            if (v.kind == Value::Number) return Value::number(-v.n);
            return binary("*", Value::number(-1.0), v);
        }
        if (tok.kind == Token::Keyword && tok.text == "not") {
            next();
            return Value::boolean(!unary().truthy());
        }
        if (tok.kind == Token::Op && tok.text == "#") {
            next();
            Value v = unary();
            if (v.kind == Value::Tab) return Value::number((double)v.t->items.size());
            if (v.kind == Value::String) return Value::number((double)v.s.size());
            fail(std::string("attempt to get length of a ") + v.type_name() + " value");
        }
        return power();
    }
    Value power() {
        Value base = suffixed(nullptr);
        if (accept_op("^")) return binary("^", base, unary());
        return base;
    }
    static std::string to_text(const Value& v) {
        if (v.kind == Value::Number) {
            if (v.n == std::floor(v.n) && std::fabs(v.n) < 1e15) return std::to_string((long long)v.n);
            std::ostringstream s;
            s.precision(17);
            s << v.n;
            return s.str();
        }
        return v.s;
    }
    Value binary(const std::string& op, const Value& a, const Value& b) {
        if (op == "and") return a.truthy() ? b : a;
        if (op == "or") return a.truthy() ? a : b;
        if (op == "==" || op == "~=") {
            bool eq = false;
            if (a.kind == b.kind) {
                switch (a.kind) {
                case Value::Nil: eq = true; break;
                case Value::Bool: eq = a.b == b.b; break;
                case Value::Number: eq = a.n == b.n; break;
                case Value::String: eq = a.s == b.s; break;
                case Value::Tab: eq = a.t == b.t; break;
                default: eq = a.s == b.s && a.t == b.t;
                }
            }
            return Value::boolean(op == "==" ? eq : !eq);
        }
        if (op == "..") {
            if ((a.kind != Value::Number && a.kind != Value::String) || (b.kind != Value::Number && b.kind != Value::String))
                fail(std::string("attempt to concatenate a ") + (a.kind == Value::Number || a.kind == Value::String ? b : a).type_name() + " value");
            return Value::string(to_text(a) + to_text(b));
        }
        if (a.kind == Value::Number && b.kind == Value::Number) {
            if (op == "+") return Value::number(a.n + b.n);
            if (op == "-") return Value::number(a.n - b.n);
            if (op == "*") return Value::number(a.n * b.n);
            if (op == "/") return Value::number(a.n / b.n);
            if (op == "%") return Value::number(a.n - std::floor(a.n / b.n) * b.n);
            if (op == "^") return Value::number(std::pow(a.n, b.n));
            if (op == "<") return Value::boolean(a.n < b.n);
            if (op == ">") return Value::boolean(a.n > b.n);
            if (op == "<=") return Value::boolean(a.n <= b.n);
            if (op == ">=") return Value::boolean(a.n >= b.n);
        }
        // expression_mt's operators (lib.lua:88-102): a tagged table on either side makes a `binary` node
        const bool a_node = a.kind == Value::Tab && !a.t->type().empty(), b_node = b.kind == Value::Tab && !b.t->type().empty();
        if ((a_node || b_node) && (op == "+" || op == "-" || op == "*" || op == "/") && (a_node || a.kind == Value::Number) && (b_node || b.kind == Value::Number)) {
            auto t = std::make_shared<Table>();
            t->set("type", Value::string("binary"));
            t->set("operator", Value::string(op == "+" ? "add" : op == "-" ? "sub" : op == "*" ? "mul" : "div"));
            t->set("lhs", a);
            t->set("rhs", b);
            return Value::table(t);
        }
        fail(std::string("attempt to perform arithmetic on a ") + a.type_name() + " and a " + b.type_name() + " value");
    }

    Value index(const Value& container, const Value& key) {
        if (container.kind != Value::Tab) fail(std::string("attempt to index a ") + container.type_name() + " value");
        if (key.kind == Value::Number) {
            const double k = key.n;
            if (k == std::floor(k) && k >= 1 && k <= (double)container.t->items.size()) return container.t->items[(size_t)k - 1];
            return Value();
        }
        if (key.kind != Value::String) return Value();
        const Value* v = container.t->find(key.s);
        if (v) return *v;
        if (!container.t->type().empty() && (key.s == "with" || key.s == "clone" || key.s == "mix")) {
            Value m;
            m.kind = Value::Method, m.s = key.s, m.t = container.t;
            return m;
        }
        return Value();
    }

    Value method(const Value& obj, const std::string& name, const std::vector<Value>& args) { // lib.lua:44-74
        if (obj.kind == Value::Tab && (name == "with" || name == "clone")) {
            auto out = std::make_shared<Table>(*obj.t); // shallow
            if (name == "with") {
                if (args.empty() || args[0].kind != Value::Tab) fail(":with expects a table");
                for (auto& kv : args[0].t->fields) out->set(kv.first, kv.second);
            }
            return Value::table(out);
        }
        if (obj.kind == Value::Tab && name == "mix" && !obj.t->type().empty()) {
            std::vector<Value> all{obj};
            all.insert(all.end(), args.begin(), args.end());
            return call_builtin("mix", all);
        }
        fail("attempt to call method '" + name + "' on a " + obj.type_name() + " value");
    }

    Value call(const Value& fn, const std::vector<Value>& args, const Target& target) {
        if (fn.kind == Value::Method) {
            std::vector<Value> rest = args;
            if (!rest.empty() && rest[0].kind == Value::Tab && rest[0].t == fn.t) rest.erase(rest.begin()); // obj.with(obj, {...})
            return method(Value::table(fn.t), fn.s, rest);
        }
        if (fn.kind == Value::Nil) fail("attempt to call a nil value" + (target.valid && !target.container ? " (global '" + target.key + "')" : std::string()));
        if (fn.kind != Value::Function) fail(std::string("attempt to call a ") + fn.type_name() + " value");
        if (fn.s == "require") return require(args);
        try {
            return call_builtin(fn.s, args);
        } catch (const LuaError& e) {
            fail(e.what());
        }
    }

    std::vector<Value> call_arguments() {
        const Token& tok = peek();
        if (tok.kind == Token::String) return {Value::string(next().text)};
        if (tok.kind == Token::Op && tok.text == "{") return {table()};
        expect_op("(");
        std::vector<Value> args;
        if (!accept_op(")")) {
            args = expression_list();
            expect_op(")");
        }
        return args;
    }

    Value primary(Target& target) {
        const Token tok = next();
        target = Target{};
        switch (tok.kind) {
        case Token::Number: return Value::number(tok.number);
        case Token::String: return Value::string(tok.text);
        case Token::Keyword:
            if (tok.text == "nil") return Value();
            if (tok.text == "true") return Value::boolean(true);
            if (tok.text == "false") return Value::boolean(false);
            --pos_;
            if (tok.text == "function") fail("function definitions are not supported in project files");
            fail("unexpected '" + tok.text + "'");
        case Token::Name: {
            target.valid = true, target.key = tok.text;
            auto l = locals_.find(tok.text);
            if (l != locals_.end()) return l->second;
            auto g = env_.globals.find(tok.text);
            return g != env_.globals.end() ? g->second : Value();
        }
        case Token::Op:
            if (tok.text == "(") {
                Value v = expression();
                expect_op(")");
                return v;
            }
            if (tok.text == "{") {
                --pos_;
                return table();
            }
            [[fallthrough]];
        default:
            --pos_;
            fail("unexpected '" + (tok.kind == Token::Eof ? std::string("end of file") : tok.text) + "'");
        }
    }

    Value suffixed(Target* target_out) {
        Target target;
        Value value = primary(target);
        for (;;) {
            const Token& tok = peek();
            if (tok.kind == Token::Op && tok.text == ".") {
                next();
                const Token key = next();
                if (key.kind != Token::Name && key.kind != Token::Keyword) fail("expected a field name");
                if (value.kind == Value::Nil) fail("attempt to index a nil value");
                Value container = value;
                value = index(container, Value::string(key.text));
                target = Target{true, container.kind == Value::Tab ? container.t : nullptr, key.text};
                if (container.kind != Value::Tab) target.valid = false;
            } else if (tok.kind == Token::Op && tok.text == "[") {
                next();
                Value key = expression();
                expect_op("]");
                if (value.kind == Value::Nil) fail("attempt to index a nil value");
                Value container = value;
                value = index(container, key);
                target = Target{key.kind == Value::String && container.kind == Value::Tab, container.t, key.s};
            } else if (tok.kind == Token::Op && tok.text == ":") {
                next();
                const std::string name = next().text;
                std::vector<Value> args = call_arguments();
                value = method(value, name, args);
                target = Target{};
            } else if ((tok.kind == Token::Op && (tok.text == "(" || tok.text == "{")) || tok.kind == Token::String) {
                std::vector<Value> args = call_arguments();
                value = call(value, args, target);
                target = Target{};
            } else {
                break;
            }
        }
        if (target_out) *target_out = target;
        return value;
    }

    Value table() {
        expect_op("{");
        auto t = std::make_shared<Table>();
        while (!accept_op("}")) {
            const Token& tok = peek();
            if (tok.kind == Token::Name && tokens_[pos_ + 1].kind == Token::Op && tokens_[pos_ + 1].text == "=") {
                const std::string key = next().text;
                next();
                t->set(key, expression());
            } else if (tok.kind == Token::Op && tok.text == "[") {
                next();
                Value key = expression();
                expect_op("]");
                expect_op("=");
                Value v = expression();
                if (key.kind == Value::String)
                    t->set(key.s, v);
                else if (key.kind == Value::Number && key.n == (double)t->items.size() + 1)
                    t->items.push_back(v);
                else
                    t->set(to_text(key), v);
            } else {
                t->items.push_back(expression());
            }
            if (!(accept_op(",") || accept_op(";"))) {
                expect_op("}");
                break;
            }
        }
        return Value::table(t);
    }

    // require: a module is another file of the same kind next to the project (mlua's package.path is the project directory)
    Value require(const std::vector<Value>& args) {
        if (args.size() != 1 || args[0].kind != Value::String) fail("require expects a module name");
        std::string rel = args[0].s;
        for (char& c : rel)
            if (c == '.') c = '/';
        const std::string path = base_dir_ + "/" + rel + ".lua";
        auto cached = env_.modules.find(path);
        if (cached != env_.modules.end()) return cached->second;
        if (loading_.count(path)) fail("circular require of '" + args[0].s + "'");
        std::ifstream f(path);
        if (!f) fail("module '" + args[0].s + "' not found (looked for " + path + ")");
        std::stringstream ss;
        ss << f.rdbuf();
        std::set<std::string> loading = loading_;
        loading.insert(path);
        const size_t slash = path.find_last_of('/');
        Value v = Evaluator(ss.str(), path.substr(slash == std::string::npos ? 0 : slash + 1), env_, base_dir_, loading).run();
        env_.modules[path] = v;
        return v;
    }
};

// ------------------------------------------------------------------------------------------------ FromLua: tagged tables -> typed tree
class Converter {
  public:
    Converter(std::string base_dir, const TextureLoader& textures) : base_dir_(std::move(base_dir)), textures_(textures) {}

    Project project(const Value& v) {
        if (v.kind != Value::Tab || !v.t->find("world")) throw ProjectError("a project file must return a table with at least `world`, `camera` and `renderer`");
        Project p;
        const Value image = v.t->get("image");
        if (image.kind == Value::Tab) {
            p.image.width = (uint32_t)number(image.t->get("width"), "image.width", 0.0);
            p.image.height = (uint32_t)number(image.t->get("height"), "image.height", 0.0);
            if (image.t->get("filter").kind != Value::Nil) p.image.filter = expression(image.t->get("filter"));
            if (image.t->get("white").kind != Value::Nil) p.image.white = expression(image.t->get("white"));
        }
        p.camera = camera(v.t->get("camera"));
        p.renderer = renderer(v.t->get("renderer"));
        const Value world = v.t->get("world");
        if (world.kind != Value::Tab) throw ProjectError("world: expected a table");
        if (world.t->get("sky").kind != Value::Nil) p.world.sky = expression(world.t->get("sky"));
        const Value objects = world.t->get("objects");
        if (objects.kind == Value::Tab)
            for (size_t i = 0; i < objects.t->items.size(); ++i) p.world.objects.push_back(object(objects.t->items[i], i));
        return p;
    }

  private:
    std::string base_dir_;
    const TextureLoader& textures_;
    std::unordered_map<const Table*, Expression> expressions_; // one node per Lua table
    std::map<std::string, Expression> texture_nodes_;          // one texture per (file, kind): project/textures.rs:56-118
    std::vector<TablePtr> keep_;

    static double number(const Value& v, const char* what, double fallback) {
        if (v.kind == Value::Nil) return fallback;
        if (v.kind != Value::Number) throw ProjectError(std::string(what) + ": expected a number");
        return v.n;
    }
    static std::optional<uint32_t> optional_uint(const Value& v, const char* what) {
        if (v.kind == Value::Nil) return std::nullopt;
        if (v.kind != Value::Number) throw ProjectError(std::string(what) + ": expected a number");
        return (uint32_t)v.n;
    }
    std::optional<Expression> optional_expression(const Value& v) {
        if (v.kind == Value::Nil) return std::nullopt;
        return expression(v);
    }

    Expression expression(const Value& v) {
        if (v.kind == Value::Number) return Expression(v.n);
        if (v.kind != Value::Tab || v.t->type().empty()) throw ProjectError(std::string("expected an expression, found a ") + v.type_name() + " value");
        auto memo = expressions_.find(v.t.get());
        if (memo != expressions_.end()) return memo->second;
        const Table& t = *v.t;
        const std::string type = t.type();
        Expression e;
        if (type == "binary") {
            const std::string op = t.get("operator").s;
            const Expression l = expression(t.get("lhs")), r = expression(t.get("rhs"));
            e = op == "add" ? l + r : op == "sub" ? l - r : op == "mul" ? l * r : l / r;
        } else if (type == "mix") {
            e = pyrite::mix(expression(t.get("lhs")), expression(t.get("rhs")), expression(t.get("amount")));
        } else if (type == "fresnel") {
            e = fresnel(expression(t.get("ior")), expression(t.get("env_ior")));
        } else if (type == "vector") {
            e = vector(expression(t.get("x")), expression(t.get("y")), expression(t.get("z")), expression(t.get("w")));
        } else if (type == "rgb") {
            e = rgb(expression(t.get("red")), expression(t.get("green")), expression(t.get("blue")));
        } else if (type == "blackbody") {
            e = blackbody(expression(t.get("temperature")));
        } else if (type == "spectrum") {
            const Value name = t.get("name");
            if (name.kind == Value::String) {
                if (name.s == "d65")
                    e = light_source::d65();
                else if (name.s == "a")
                    e = light_source::a();
                else
                    throw ProjectError("unknown builtin spectrum: " + name.s);
            } else {
                const Value points = t.get("points");
                if (points.kind != Value::Tab) throw ProjectError("spectrum: expected `points`");
                const std::string format = t.get("format").kind == Value::String ? t.get("format").s : "array";
                if (format == "array") {
                    std::vector<float> values;
                    for (const Value& p : points.t->items) values.push_back((float)number(p, "spectrum point", 0.0));
                    e = spectrum_array((float)number(t.get("min"), "spectrum.min", 0.0), (float)number(t.get("max"), "spectrum.max", 0.0), std::move(values));
                } else if (format == "curve") {
                    std::vector<std::pair<float, float>> pairs;
                    for (const Value& p : points.t->items) {
                        if (p.kind != Value::Tab || p.t->items.size() != 2) throw ProjectError("spectrum curve: expected {wavelength, value} pairs");
                        pairs.emplace_back((float)number(p.t->items[0], "curve point", 0.0), (float)number(p.t->items[1], "curve point", 0.0));
                    }
                    e = spectrum_curve(std::move(pairs));
                } else {
                    throw ProjectError("unknown spectrum format '" + format + "'");
                }
            }
        } else if (type == "color_texture" || type == "mono_texture") {
            const bool mono = type == "mono_texture", linear = t.get("linear").truthy();
            std::string path = t.get("path").s;
            if (path.empty() || path[0] != '/') path = base_dir_ + "/" + path;
            const std::string key = path + (linear ? "|linear" : "|srgb") + (mono ? "|mono" : "|color");
            auto known = texture_nodes_.find(key);
            if (known != texture_nodes_.end()) {
                e = known->second;
            } else {
                uint32_t width = 0, height = 0;
                std::vector<float> texels = textures_ ? textures_(path, linear, mono, width, height) : load_texture_file(path, linear, mono, width, height);
                e = mono ? mono_texture(width, height, std::move(texels)) : color_texture(width, height, std::move(texels));
                texture_nodes_.emplace(key, e);
            }
        } else {
            throw ProjectError("not an expression: a `" + type + "` table");
        }
        expressions_.emplace(v.t.get(), e);
        keep_.push_back(v.t);
        return e;
    }

    SurfaceMaterial surface(const Value& v) { // project/materials.rs:5-35
        if (v.kind != Value::Tab) throw ProjectError("missing material");
        const Table& t = *v.t;
        const std::string type = t.type();
        if (type == "diffuse") return material::diffuse(expression(t.get("color")));
        if (type == "emissive") return material::emissive(expression(t.get("color")));
        if (type == "mirror") return material::mirror(expression(t.get("color")));
        if (type == "refractive")
            return material::refractive(expression(t.get("color")), expression(t.get("ior")), optional_expression(t.get("dispersion")), optional_expression(t.get("env_ior")),
                                        optional_expression(t.get("env_dispersion")));
        if (type == "mix") return pyrite::mix(surface(t.get("lhs")), surface(t.get("rhs")), expression(t.get("amount")));
        if (type == "binary") {
            if (t.get("operator").s != "add") throw ProjectError("materials can only be added");
            return surface(t.get("lhs")) + surface(t.get("rhs"));
        }
        throw ProjectError("unknown material type " + type);
    }
    Material material_of(const Value& v) {
        if (v.kind != Value::Tab) throw ProjectError("missing material");
        Material m(surface(v.t->get("surface")));
        if (v.t->get("normal_map").kind != Value::Nil) m.normal_map = expression(v.t->get("normal_map"));
        return m;
    }
    LookAt look_at(const Value& v) {
        if (v.kind != Value::Tab || v.t->type() != "look_at") throw ProjectError("unknown transform");
        LookAt t;
        t.from = v.t->get("from").kind == Value::Nil ? Expression(0.0) : expression(v.t->get("from"));
        t.to = v.t->get("to").kind == Value::Nil ? Expression(0.0) : expression(v.t->get("to"));
        t.up = optional_expression(v.t->get("up"));
        return t;
    }
    CameraProject camera(const Value& v) {
        if (v.kind != Value::Tab || v.t->type() != "perspective") throw ProjectError("unknown camera");
        CameraProject c;
        c.transform = look_at(v.t->get("transform"));
        c.fov = expression(v.t->get("fov"));
        c.focus_distance = optional_expression(v.t->get("focus_distance"));
        c.aperture = optional_expression(v.t->get("aperture"));
        return c;
    }
    RendererProject renderer(const Value& v) {
        if (v.kind != Value::Tab) throw ProjectError("missing renderer");
        if (v.t->type() != "simple") throw ProjectError("renderer." + v.t->type() + " is out of scope: only the camera-to-light `simple` renderer is built");
        RendererProject r;
        r.pixel_samples = (uint32_t)number(v.t->get("pixel_samples"), "renderer.pixel_samples", 1.0);
        r.bounces = optional_uint(v.t->get("bounces"), "renderer.bounces");
        r.light_samples = optional_uint(v.t->get("light_samples"), "renderer.light_samples");
        r.spectrum_samples = optional_uint(v.t->get("spectrum_samples"), "renderer.spectrum_samples");
        r.spectrum_resolution = optional_uint(v.t->get("spectrum_resolution"), "renderer.spectrum_resolution");
        r.tile_size = optional_uint(v.t->get("tile_size"), "renderer.tile_size");
        return r;
    }
    WorldObject object(const Value& v, size_t index) {
        if (v.kind != Value::Tab) throw ProjectError("objects[" + std::to_string(index) + "]: expected an object");
        const Table& t = *v.t;
        const std::string type = t.type();
        if (type == "sphere") return Sphere{expression(t.get("position")), expression(t.get("radius")), material_of(t.get("material")), optional_expression(t.get("texture_scale"))};
        if (type == "plane") return Plane{expression(t.get("origin")), expression(t.get("normal")), material_of(t.get("material")), optional_expression(t.get("texture_scale"))};
        if (type == "mesh") {
            Mesh m;
            if (t.get("file").kind != Value::String) throw ProjectError("objects[" + std::to_string(index) + "]: mesh.file: expected a path");
            m.file = t.get("file").s;
            const Value materials = t.get("materials");
            if (materials.kind == Value::Tab)
                for (auto& kv : materials.t->fields) m.materials[kv.first] = material_of(kv.second);
            m.scale = optional_expression(t.get("scale"));
            if (t.get("transform").kind != Value::Nil) m.transform = look_at(t.get("transform"));
            return m;
        }
        if (type == "directional_light") return DirectionalLight{expression(t.get("direction")), expression(t.get("width")), expression(t.get("color"))};
        if (type == "point_light") return PointLight{expression(t.get("position")), expression(t.get("color"))};
        if (type == "ray_marched") throw ProjectError("ray-marched shapes are out of scope for the GPU path (SURVEY.md section 8)");
        throw ProjectError("objects[" + std::to_string(index) + "]: unknown object type " + type);
    }
};

} // namespace

LoadedProject evaluate_project(const std::string& text, const std::string& name, const std::string& base_dir, const TextureLoader& textures) {
    Environment env;
    const Value v = Evaluator(text, name, env, base_dir, {}).run();
    LoadedProject out;
    out.base_dir = base_dir;
    out.project = Converter(base_dir, textures).project(v);
    return out;
}

LoadedProject load_project(const std::string& path, const TextureLoader& textures) { // main.rs:111-134 parse_project
    std::ifstream f(path);
    if (!f) throw ProjectError("could not open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const size_t slash = path.find_last_of('/');
    const std::string base_dir = slash == std::string::npos ? "." : path.substr(0, slash);
    return evaluate_project(ss.str(), path.substr(slash == std::string::npos ? 0 : slash + 1), base_dir, textures);
}

} // namespace pyrite

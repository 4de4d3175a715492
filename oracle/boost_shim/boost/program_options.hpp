// Stand-in for boost/program_options.hpp (Boost is absent from this image and from the reference's redist/): the
// subset the reference's option parsers use, so that its OWN command-line front ends
// (L/starling_common/starling_base_option_parser.cpp, L/applications/{starling,strelka}/*_option_parser.cpp) compile
// and the test binaries under oracle/_ref/ take the same argv the pyflow workflow passes to starling2 / strelka2.
//
// TEST INFRASTRUCTURE ONLY.  Semantics kept: long options `--name value` / `--name=value`, `name,x` short aliases,
// default_value / implicit_value / zero_tokens / multitoken / composing, vector<T> options composing over repeated
// occurrences, multiple occurrences of a scalar option being an error, user `validate(any&, tokens, T*, int)`
// overloads found by ADL (blt_util/PrettyFloat.hh), store/notify into bound variables.  Not kept: prefix guessing,
// positional options, config files, environment.
#pragma once
#include <limits>
#include "boost/utility.hpp"
#include "boost/lexical_cast.hpp"
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>
namespace boost {
class bad_any_cast : public std::bad_cast {
public:
    const char* what() const noexcept override { return "boost::bad_any_cast"; }
};
class any {
    struct holder_base { virtual ~holder_base() {} virtual holder_base* clone() const = 0; virtual const std::type_info& type() const = 0; };
    template <typename T> struct holder : holder_base {
        T held; explicit holder(const T& t) : held(t) {}
        holder_base* clone() const override { return new holder(held); }
        const std::type_info& type() const override { return typeid(T); }
    };
    std::unique_ptr<holder_base> _p;
public:
    any() {}
    template <typename T> any(const T& t) : _p(new holder<T>(t)) {}
    any(const any& o) : _p(o._p ? o._p->clone() : nullptr) {}
    any& operator=(const any& o) { _p.reset(o._p ? o._p->clone() : nullptr); return *this; }
    template <typename T> any& operator=(const T& t) { _p.reset(new holder<T>(t)); return *this; }
    bool empty() const { return !_p; }
    template <typename T> friend T* any_cast(any* a);
    template <typename T> friend const T* any_cast(const any* a);
};
template <typename T> T* any_cast(any* a) {
    if (!a || !a->_p || a->_p->type() != typeid(T)) return nullptr;
    return &static_cast<any::holder<T>*>(a->_p.get())->held;
}
template <typename T> const T* any_cast(const any* a) {
    if (!a || !a->_p || a->_p->type() != typeid(T)) return nullptr;
    return &static_cast<const any::holder<T>*>(a->_p.get())->held;
}
template <typename T> T any_cast(const any& a) {
    const T* p(any_cast<T>(&a));
    if (!p) throw bad_any_cast();
    return *p;
}
template <typename T> T any_cast(any& a) {
    T* p(any_cast<T>(&a));
    if (!p) throw bad_any_cast();
    return *p;
}

namespace program_options {

class error : public std::logic_error {
public:
    explicit error(const std::string& w) : std::logic_error(w) {}
};
struct validation_error : error {
    enum kind_t { multiple_values_not_allowed = 30, at_least_one_value_required, invalid_bool_value, invalid_option_value, invalid_option };
    explicit validation_error(kind_t k, const std::string& value = std::string())
        : error(std::string(k == invalid_bool_value ? "invalid bool value" : k == multiple_values_not_allowed ? "multiple values not allowed"
                            : k == at_least_one_value_required ? "at least one value required" : "invalid option value") +
                (value.empty() ? "" : " '" + value + "'")) {}
};
struct unknown_option : error { explicit unknown_option(const std::string& n) : error("unrecognised option '" + n + "'") {} };
struct multiple_occurrences : error { explicit multiple_occurrences(const std::string& n) : error("option '" + n + "' cannot be specified more than once") {} };
struct invalid_command_line_syntax : error { explicit invalid_command_line_syntax(const std::string& w) : error(w) {} };

namespace validators {
inline const std::string& get_single_string(const std::vector<std::string>& v, bool allow_empty = false) {
    static const std::string empty;
    if (v.size() > 1) throw validation_error(validation_error::multiple_values_not_allowed);
    if (v.size() == 1) return v.front();
    if (!allow_empty) throw validation_error(validation_error::at_least_one_value_required);
    return empty;
}
inline void check_first_occurrence(const any& v) {
    if (!v.empty()) throw multiple_occurrences("");
}
}

// generic validators; `long` last parameter so that a user overload taking `int` is preferred (Boost's own trick)
template <typename T> void validate(any& v, const std::vector<std::string>& xs, T*, long) {
    validators::check_first_occurrence(v);
    const std::string& s(validators::get_single_string(xs));
    try { v = any(lexical_cast<T>(s)); }
    catch (const bad_lexical_cast&) { throw validation_error(validation_error::invalid_option_value, s); }
}
inline void validate(any& v, const std::vector<std::string>& xs, bool*, int) {
    validators::check_first_occurrence(v);
    std::string s(validators::get_single_string(xs, true));
    for (auto& c : s) c = static_cast<char>(std::tolower(c));
    if (s.empty() || s == "on" || s == "yes" || s == "1" || s == "true") v = any(true);
    else if (s == "off" || s == "no" || s == "0" || s == "false") v = any(false);
    else throw validation_error(validation_error::invalid_bool_value, s);
}
inline void validate(any& v, const std::vector<std::string>& xs, std::string*, int) {
    validators::check_first_occurrence(v);
    v = any(validators::get_single_string(xs));
}
template <typename T> void validate(any& v, const std::vector<std::string>& xs, std::vector<T>*, int) {
    if (v.empty()) v = any(std::vector<T>());
    std::vector<T>* tv(any_cast<std::vector<T>>(&v));
    for (const auto& x : xs) {
        any a;
        std::vector<std::string> one(1, x);
        validate(a, one, (T*)0, 0);
        tv->push_back(any_cast<T>(a));
    }
}

class value_semantic {
public:
    virtual ~value_semantic() {}
    virtual unsigned min_tokens() const = 0;
    virtual unsigned max_tokens() const = 0;
    virtual bool is_composing() const = 0;
    virtual void parse(any& value_store, const std::vector<std::string>& new_tokens) const = 0;
    virtual bool apply_default(any& value_store) const = 0;
    virtual void notify(const any& value_store) const = 0;
    virtual std::string name() const = 0;
};

class untyped_value : public value_semantic {
public:
    unsigned min_tokens() const override { return 0; }
    unsigned max_tokens() const override { return 0; }
    bool is_composing() const override { return false; }
    void parse(any& value_store, const std::vector<std::string>&) const override {
        if (!value_store.empty()) throw multiple_occurrences("");
        value_store = any(std::string());
    }
    bool apply_default(any&) const override { return false; }
    void notify(const any&) const override {}
    std::string name() const override { return ""; }
};

template <typename T> struct is_vector_type { enum { value = 0 }; };
template <typename T> struct is_vector_type<std::vector<T>> { enum { value = 1 }; };

template <typename T> class typed_value : public value_semantic {
public:
    explicit typed_value(T* store_to) : _store_to(store_to) {}
    typed_value* default_value(const T& v) { _default = any(v); return this; }
    typed_value* default_value(const T& v, const std::string&) { _default = any(v); return this; }
    typed_value* implicit_value(const T& v) { _implicit = any(v); return this; }
    typed_value* multitoken() { _multitoken = true; return this; }
    typed_value* composing() { _composing = true; return this; }
    typed_value* zero_tokens() { _zero_tokens = true; return this; }
    typed_value* required() { return this; }

    unsigned min_tokens() const override { return (_zero_tokens || !_implicit.empty()) ? 0 : 1; }
    unsigned max_tokens() const override { return _multitoken ? 32000 : (_zero_tokens ? 0 : 1); }
    bool is_composing() const override { return _composing || is_vector_type<T>::value; }
    void parse(any& value_store, const std::vector<std::string>& new_tokens) const override {
        if (new_tokens.empty() && !_implicit.empty()) value_store = _implicit;
        else validate(value_store, new_tokens, (T*)0, 0);
    }
    bool apply_default(any& value_store) const override {
        if (_default.empty()) return false;
        value_store = _default;
        return true;
    }
    void notify(const any& value_store) const override {
        const T* v(any_cast<T>(&value_store));
        if (_store_to && v) *_store_to = *v;
    }
    std::string name() const override { return "arg"; }
private:
    T* _store_to;
    any _default, _implicit;
    bool _multitoken = false, _composing = false, _zero_tokens = false;
};

template <typename T> typed_value<T>* value() { return new typed_value<T>(nullptr); }
template <typename T> typed_value<T>* value(T* v) { return new typed_value<T>(v); }
inline typed_value<bool>* bool_switch(bool* v = nullptr) { auto* r(new typed_value<bool>(v)); r->default_value(false); r->zero_tokens(); return r; }

struct option_description {
    std::string long_name, short_name, description;
    std::shared_ptr<const value_semantic> semantic;
};

class options_description;
class options_description_easy_init {
public:
    explicit options_description_easy_init(options_description* owner) : _owner(owner) {}
    options_description_easy_init& operator()(const char* name, const char* description);
    options_description_easy_init& operator()(const char* name, const value_semantic* s);
    options_description_easy_init& operator()(const char* name, const value_semantic* s, const char* description);
private:
    options_description* _owner;
};

class options_description {
public:
    options_description() {}
    explicit options_description(const std::string& caption) : _caption(caption) {}
    options_description_easy_init add_options() { return options_description_easy_init(this); }
    options_description& add(const options_description& d) {
        _groups.push_back(d);
        return *this;
    }
    void add_one(const char* name, const value_semantic* s, const char* description) {
        option_description od;
        const std::string n(name);
        const auto comma(n.find(','));
        od.long_name = n.substr(0, comma);
        if (comma != std::string::npos) od.short_name = n.substr(comma + 1);
        od.description = description ? description : "";
        od.semantic.reset(s);
        _options.push_back(od);
    }
    void collect(std::vector<const option_description*>& out) const {
        for (const auto& o : _options) out.push_back(&o);
        for (const auto& g : _groups) g.collect(out);
    }
    void print(std::ostream& os) const {
        if (!_caption.empty()) os << _caption << ":\n";
        for (const auto& o : _options) {
            std::string head("  ");
            if (!o.short_name.empty()) head += "-" + o.short_name + " [ --" + o.long_name + " ]";
            else head += "--" + o.long_name;
            const std::string arg(o.semantic->name());
            if (!arg.empty() && o.semantic->max_tokens() > 0) head += " " + arg;
            os << head;
            if (head.size() < 38) os << std::string(38 - head.size(), ' ');
            else os << "\n" << std::string(38, ' ');
            os << o.description << "\n";
        }
        for (const auto& g : _groups) { os << "\n"; g.print(os); }
    }
private:
    std::string _caption;
    std::vector<option_description> _options;
    std::vector<options_description> _groups;
};
inline std::ostream& operator<<(std::ostream& os, const options_description& d) { d.print(os); return os; }

inline options_description_easy_init& options_description_easy_init::operator()(const char* name, const char* description) {
    _owner->add_one(name, new untyped_value(), description);
    return *this;
}
inline options_description_easy_init& options_description_easy_init::operator()(const char* name, const value_semantic* s) {
    _owner->add_one(name, s, "");
    return *this;
}
inline options_description_easy_init& options_description_easy_init::operator()(const char* name, const value_semantic* s, const char* description) {
    _owner->add_one(name, s, description);
    return *this;
}

struct basic_option {
    std::string string_key;
    std::vector<std::string> value;
};
struct parsed_options {
    std::vector<basic_option> options;
    const options_description* description = nullptr;
};

class command_line_parser {
public:
    command_line_parser(int argc, const char* const argv[]) { for (int i(1); i < argc; ++i) _args.push_back(argv[i]); }
    explicit command_line_parser(const std::vector<std::string>& args) : _args(args) {}
    command_line_parser& options(const options_description& d) { _desc = &d; return *this; }
    parsed_options run() {
        parsed_options result;
        result.description = _desc;
        std::vector<const option_description*> all;
        if (_desc) _desc->collect(all);
        auto find_long = [&](const std::string& n) -> const option_description* {
            for (const auto* o : all) if (o->long_name == n) return o;
            return nullptr;
        };
        auto find_short = [&](const std::string& n) -> const option_description* {
            for (const auto* o : all) if (o->short_name == n) return o;
            return nullptr;
        };
        auto is_option_token = [&](const std::string& t) {
            if (t.size() >= 3 && t[0] == '-' && t[1] == '-') return true;
            if (t.size() == 2 && t[0] == '-' && find_short(t.substr(1))) return true;
            return false;
        };
        for (size_t i(0); i < _args.size();) {
            const std::string& tok(_args[i]);
            const option_description* od(nullptr);
            basic_option bo;
            bool has_adjacent(false);
            if (tok.size() >= 3 && tok[0] == '-' && tok[1] == '-') {
                std::string name(tok.substr(2));
                const auto eq(name.find('='));
                if (eq != std::string::npos) {
                    bo.value.push_back(name.substr(eq + 1));
                    name = name.substr(0, eq);
                    has_adjacent = true;
                }
                od = find_long(name);
                if (!od) throw unknown_option(tok);
            } else if (tok.size() == 2 && tok[0] == '-') {
                od = find_short(tok.substr(1));
                if (!od) throw unknown_option(tok);
            } else {
                throw invalid_command_line_syntax("unexpected positional argument '" + tok + "'");
            }
            bo.string_key = od->long_name;
            ++i;
            const unsigned min_t(od->semantic->min_tokens()), max_t(od->semantic->max_tokens());
            if (has_adjacent && max_t == 0) throw invalid_command_line_syntax("option '--" + od->long_name + "' does not take any arguments");
            while (bo.value.size() < max_t && i < _args.size() && !is_option_token(_args[i])) {
                bo.value.push_back(_args[i]);
                ++i;
            }
            if (bo.value.size() < min_t) throw invalid_command_line_syntax("the required argument for option '--" + od->long_name + "' is missing");
            result.options.push_back(bo);
        }
        return result;
    }
private:
    std::vector<std::string> _args;
    const options_description* _desc = nullptr;
};
inline parsed_options parse_command_line(int argc, const char* const argv[], const options_description& d) {
    return command_line_parser(argc, argv).options(d).run();
}

class variable_value {
public:
    variable_value() {}
    variable_value(const any& v, bool defaulted) : _v(v), _defaulted(defaulted) {}
    template <typename T> const T& as() const {
        const T* p(any_cast<T>(&_v));
        if (!p) throw bad_any_cast();
        return *p;
    }
    bool empty() const { return _v.empty(); }
    bool defaulted() const { return _defaulted; }
    const any& value() const { return _v; }
    any& value() { return _v; }
    std::shared_ptr<const value_semantic> semantic;
private:
    any _v;
    bool _defaulted = false;
};

class variables_map : public std::map<std::string, variable_value> {
public:
    size_t count(const std::string& k) const { return std::map<std::string, variable_value>::count(k); }
    const variable_value& operator[](const std::string& k) const {
        static const variable_value empty;
        const auto it(find(k));
        return it == end() ? empty : it->second;
    }
    variable_value& slot(const std::string& k) { return std::map<std::string, variable_value>::operator[](k); }
};

inline void store(const parsed_options& po, variables_map& vm) {
    std::vector<const option_description*> all;
    if (po.description) po.description->collect(all);
    std::map<std::string, any> fresh; // values parsed in THIS store call
    for (const auto& bo : po.options) {
        const option_description* od(nullptr);
        for (const auto* o : all) if (o->long_name == bo.string_key) od = o;
        if (!od) continue;
        // a value stored by an earlier store() wins unless it was only a default
        const auto prior(vm.find(bo.string_key));
        if (prior != vm.end() && !prior->second.defaulted() && !fresh.count(bo.string_key)) continue;
        any& v(fresh[bo.string_key]);
        try { od->semantic->parse(v, bo.value); }
        catch (const multiple_occurrences&) { throw multiple_occurrences("--" + bo.string_key); }
        catch (const validation_error& e) { throw error(std::string("the argument for option '--") + bo.string_key + "' is invalid: " + e.what()); }
    }
    for (const auto& kv : fresh) {
        const option_description* od(nullptr);
        for (const auto* o : all) if (o->long_name == kv.first) od = o;
        variable_value vv(kv.second, false);
        vv.semantic = od->semantic;
        vm.slot(kv.first) = vv;
    }
    for (const auto* o : all) {
        if (vm.count(o->long_name)) continue;
        any def;
        if (o->semantic->apply_default(def)) {
            variable_value vv(def, true);
            vv.semantic = o->semantic;
            vm.slot(o->long_name) = vv;
        }
    }
}
inline void notify(variables_map& vm) {
    for (auto& kv : vm) if (kv.second.semantic) kv.second.semantic->notify(kv.second.value());
}

}}

// Forward declaration for the `friend struct sk_adapter::Access;` line that adapter/apply_hooks.py adds to the position
// processor classes of the reference (no data member is added: the classes keep their layout, so the reference's other
// translation units link against the hooked ones unchanged).
#pragma once
namespace sk_adapter
{
struct Access;
struct GvcfAccess; // (sk_adapter_gvcf.cpp: the gVCF writer's pipe)
}

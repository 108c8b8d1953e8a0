package org.apache.singa;

import static org.junit.Assert.assertArrayEquals;
import static org.junit.Assert.assertEquals;

import org.junit.Test;

/** The tensor checks of the reference's Java test, plus a product and a reduction. */
public class TestTensor {

    @Test
    public void testTensorFunc() {
        long[] s = {2, 3};
        try (Tensor t1 = new Tensor(s)) {
            t1.SetFloatValue(0.1f);
            float[] data = new float[6];
            try (Tensor t2 = Singa.Square(t1)) {
                t2.GetFloatValue(data, 6);
                for (int i = 0; i < 6; i++) assertEquals(0.01, data[i], 1e-4);
            }
            for (int i = 0; i < 6; i++) data[i] = i * 1.0f;
            try (Tensor t3 = new Tensor(s)) {
                t3.CopyFloatDataFromHostPtr(data, 6);
                float[] back = new float[6];
                t3.GetFloatValue(back, 6);
                assertArrayEquals(data, back, 1e-6f);
                try (Tensor tt = t3.T(); Tensor p = Singa.Mult(t3, tt); Tensor r = Singa.Sum(p, 0)) {
                    assertArrayEquals(new long[] {2, 2}, p.shape());
                    float[] pv = new float[4];
                    p.GetFloatValue(pv, 4);
                    assertArrayEquals(new float[] {5, 14, 14, 50}, pv, 1e-5f);
                    float[] rv = new float[2];
                    r.GetFloatValue(rv, 2);
                    assertArrayEquals(new float[] {19, 64}, rv, 1e-5f);
                }
            }
        }
    }
}
